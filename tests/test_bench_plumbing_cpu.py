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
