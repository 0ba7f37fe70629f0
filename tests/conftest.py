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


def pytest_sessionstart(session):
    """The oracle is a Python loop over small fp64 matrix products; on a many-core host (the GPU box has 128 threads) torch's
    intra-op pool turns every one of them into a 128-way fork / join and the -m gpu suite spends 10 x longer in the oracle than
    on this 8-core container (measured: 505 s for five tests).  A handful of threads is the fastest setting on both."""
    try:
        import torch
        torch.set_num_threads(max(1, min(4, os.cpu_count() or 1)))
    except Exception:
        pass
