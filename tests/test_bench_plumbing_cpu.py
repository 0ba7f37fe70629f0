"""Host-side plumbing of bench.py that the driver's JSON line depends on (no GPU): the lookup of the committed PMC traffic
summaries for `roofline.traffic`, and the workspace size the C ABI reports for the TTT-MLP backward (two slot buffers)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_traffic_lookup_matches_geometry_and_prefers_the_newest_summary():
    import bench
    by, src = bench.pmc_traffic("ttt_mlp_bwd_scan[mfma]", 1, 48, 804)
    assert src and os.path.exists(os.path.join(ROOT, src)), src
    d = json.load(open(os.path.join(ROOT, src)))
    assert (d["geometry"]["B"], d["geometry"]["NH"], d["geometry"]["NC"]) == (1, 48, 804)
    assert by == d["kernels"]["ttt_mlp_bwd_scan[mfma]"]["traffic_bytes_per_backward"]
    # every committed summary of that geometry sorts at or below the chosen one (newest round / call wins)
    import glob
    same = [f for f in glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))
            if json.load(open(f)).get("geometry", {"B": 1, "NH": 48, "NC": 282}).get("NC") == 804]
    assert os.path.join(ROOT, src) == sorted(same)[-1]
    # a geometry nobody measured: no number is better than a wrong one
    assert bench.pmc_traffic("ttt_mlp_bwd_scan[mfma]", 3, 48, 777) == (None, None)
    # algorithmic bytes stay far below the measured traffic (the slot round trip): the ratio DESIGN.md quotes
    k = d["kernels"]["ttt_mlp_bwd_scan[mfma]"]
    assert 15 < by / k["algorithmic_bytes"] < 40


def test_mlp_backward_workspace_holds_two_slot_buffers():
    """Revision 4 of the TTT-MLP backward: a step record is 7 fragment arrays x 4 hidden slices x 8 KiB + 48.5 KiB of owner
    rows + the 8-KiB gZ2 tile = 280.5 KiB (round 2: 570 KiB), of which the recompute kernel writes 120.5 KiB."""
    import test_time_training as ext
    lib = ext.load_library()
    import torch
    dims = ext._dims(1, 48, 804, 64, 64, 16, torch.bfloat16)
    lib.ttt_hip_mlp_backward_workspace.restype = ctypes.c_size_t
    ws = lib.ttt_hip_mlp_backward_workspace(ctypes.byref(dims))
    slot = 4 * 7 * 8 * 1024 + 3 * 64 * 64 * 4 + 64 * 8 + 64 * 64 * 2
    assert slot == 287232
    steps = 5 * 16 + 1                             # 5 checkpoint groups per chunk at 48 heads + the post-update slot
    assert 2 * 48 * steps * slot < ws < 2 * 48 * steps * slot + (64 << 20), ws


def test_counter_summaries_reproduce_from_the_committed_csvs():
    """tools/wait_lds.py and tools/mfma_util.py on the rocprofv3 --pmc CSVs under profiles/: the summaries quoted in DESIGN.md
    (section 8 table, section 4 MFMA utilisation) follow from the committed counter files."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wait_lds.py"), os.path.join(ROOT, "profiles", "r3p_op_nc804_pmc_wait_lds.csv")],
                         capture_output=True, text=True, check=True).stdout
    rows = {l.split()[0]: l.split() for l in out.splitlines()[1:]}
    sweep = next(v for k, v in rows.items() if "mlp_bwd_cluster4_kernel" in k)
    scan = next(v for k, v in rows.items() if "mlp_scan8_kernel" in k)
    pct = lambda s: float(s.rstrip("%"))
    assert abs(pct(sweep[3]) - 72.7) < 0.2            # parked share of the sweep's wave cycles
    assert abs(pct(scan[-1]) - 19.9) < 0.2            # bank-conflict share of the forward scan's CU cycles
    committed = open(os.path.join(ROOT, "profiles", "r3p_wait_lds_summary.txt")).read()
    assert out.strip() in committed
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_util.py"), os.path.join(ROOT, "profiles", "r3j_op_nc804_pmc_sq.csv")],
                         capture_output=True, text=True, check=True).stdout
    assert "mlp_bwd_cluster4_kernel" in out and "mlp_scan8_kernel" in out
