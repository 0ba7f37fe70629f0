"""pytest config: registers the `gpu` marker and puts the package dir on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ttt-video-dit_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
