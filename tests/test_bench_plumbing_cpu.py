"""Host-side plumbing of bench.py that the driver's JSON line depends on (no GPU): the lookup of the committed PMC traffic
summaries for `roofline.traffic`, and the workspace size the C ABI reports for the TTT-MLP backward (two slot buffers)."""
import ctypes
import json
import os
import sys

import pytest

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
    assert 8 < by / k["algorithmic_bytes"] < 40          # (round 3: x15.7, round 4: x11.6)


def test_mlp_backward_workspace_holds_three_slot_buffers():
    """Revision 4 of the TTT-MLP backward: a step record is 5 fragment arrays (Z1, Z1b, dZ1, dZ1b, gZ1) x 4 hidden slices x 8 KiB + 48.5 KiB
    of owner rows + the 8-KiB gZ2 tile = 216.5 KiB (round 2: 570 KiB; rounds 3 - 5: 280.5 with the per-step dW1' / W1 images the
    group-sequential tail of round 6 no longer needs), of which the recompute kernel writes 104.5 KiB.  Round 6 (schedule 2: the
    recompute of chunk c - 1, the sweep of chunk c and the tail of chunk c + 1 run at the same time): THREE record buffers, two under
    the older schedules (debug option overlap_tail 0 / 1); plus one fp32 dW1 anchor (64 KiB) per (b, h, checkpoint group)."""
    import test_time_training as ext
    lib = ext.load_library()
    import torch
    dims = ext._dims(1, 48, 804, 64, 64, 16, torch.bfloat16)
    lib.ttt_hip_mlp_backward_workspace.restype = ctypes.c_size_t
    ws = lib.ttt_hip_mlp_backward_workspace(ctypes.byref(dims))
    slot = 4 * 5 * 8 * 1024 + 3 * 64 * 64 * 4 + 64 * 8 + 64 * 64 * 2
    assert slot == 221696
    steps = 5 * 16 + 1                             # 5 checkpoint groups per chunk at 48 heads + the post-update slot
    anchors = 48 * 51 * 64 * 256 * 4               # K = ceil(804 / 16) = 51 groups
    assert 3 * 48 * steps * slot + anchors < ws < 3 * 48 * steps * slot + anchors + (64 << 20), ws
    ext.debug_option("overlap_tail", 1)
    try:
        ws1 = lib.ttt_hip_mlp_backward_workspace(ctypes.byref(dims))
    finally:
        ext.debug_option("overlap_tail", 2)
    assert 2 * 48 * steps * slot + anchors < ws1 < 2 * 48 * steps * slot + anchors + (64 << 20), ws1


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


def test_forward_scan_lds_model_is_near_the_device_counter():
    """tools/lds_bank_model.py enumerates the LDS instructions of one forward-scan step (csrc/ttt_mfma2.hip) under the bank
    model: its conflict share of the LDS passes (35.5 %) against SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of `mlp_scan8_kernel`
    on the device (38.3 %, profiles/r3p_wait_lds_summary.txt: 19.9 of 51.8 points); the swizzled layout it proposes removes every
    tile conflict (what remains are the owners' fp32 partial-row reads)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(ROOT, "tools", "lds_bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)

    def share(lay):
        P = C = 0
        by = {}
        for cls, nbytes, write, addr in m.scan_step(lay):
            p, c = m.bank_cost(addr, nbytes, write)
            P += p; C += c
            by[cls] = by.get(cls, 0) + c
        return P, C, by

    P, C, by = share(m.Layout(72, False))
    assert abs(C / P - 19.9 / 51.8) < 0.04, C / P
    P2, C2, by2 = share(m.Layout(72, True))
    assert P2 < 0.75 * P
    assert all(v == 0 for k, v in by2.items() if k.startswith(("pi_read", "tr ", "st_image", "tile park")))
    # the variant the kernel carries (always on since round 5): the row walkers' conflicts go, the transposed reads' stay
    P3, C3, by3 = share(m.Layout(72, False, True))
    assert (P3, C3) == (P - 1280, C - 1280)
    assert all(v == 0 for k, v in by3.items() if k.startswith(("pi_read", "st_image", "tile park")))
    assert all(by3[k] == by[k] for k in by if k.startswith("tr "))


_LINE = {"metric": "DiT+TTT fwd/bwd video-tokens/sec", "value": 7000.0, "unit": "video-tokens/s", "n_gpus": 1, "steps": 7, "warmup": 2,
         "ms_per_step": 7000.0, "config": {"workload": "w", "remat_free_layers": 13, "remat_keep": ["attn"], "valid": True},
         "roofline": {"kernel": "ttt_mlp_bwd_scan[mfma]", "avg_launch_ms": 10.9, "frac": 0.04,
                      "other": {"fwd": {"avg_ms": 6.0}, "attn_fwd": {"avg_ms": 4.4}, "attn_bwd": {"avg_ms": 13.0}}},
         "peak_mem_gib": 236.0, "peak_reserved_gib": 245.0}


def _orchestrate(monkeypatch, capsys, argv, replies, host_gib=2000.0):
    """bench.main() as the driver starts it (no launcher environment) with run_child replaced by a script of replies;
    returns (the commands it ran, the JSON objects it printed)"""
    import copy
    import bench
    calls = []

    def fake_run_child(cmd, timeout, env=None):
        calls.append((list(cmd), dict(env) if env else None))
        r = replies[min(len(calls) - 1, len(replies) - 1)]
        return r[0], copy.deepcopy(r[1]), list(r[2])

    monkeypatch.setattr(bench, "run_child", fake_run_child)
    monkeypatch.setattr(bench, "host_room_gib", lambda: host_gib)          # (a GPU box has 3 TB of host memory; this container 64 GB)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NCCL_DEBUG"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    try:
        bench.main()
        rc = 0
    except SystemExit as ex:
        rc = ex.code
    out = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    return calls, out, rc


def test_bare_multi_gpu_invocation_starts_the_ranks_as_children(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher environment (how the driver starts every run): a thin parent starts one rank
    per GPU under torch.distributed.run as a CHILD (rendezvous on 127.0.0.1, its own arguments handed through, RCCL's INIT / GRAPH
    decisions logged to per-rank files), and prints the child's one JSON line.  Reference: scripts/train_singlenode.sh:25-38."""
    import bench
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "4", "--steps", "7", "--warmup", "2"], [(0, dict(_LINE, n_gpus=4), [])])
    assert rc == 0 and len(calls) == 1 and len(out) == 1 and out[0]["value"] == 7000.0 and out[0]["n_gpus"] == 4
    argv, env = calls[0]
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    i = argv.index(os.path.abspath(bench.__file__))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--role", "worker"]
    assert env["NCCL_DEBUG"] == "INFO" and "rccl." in env["NCCL_DEBUG_FILE"]
    assert "ctx3s" not in out[0] and "cpu_baseline" not in out[0]           # legs and the CPU baseline are N = 1 only


def test_failed_multi_gpu_attempt_is_retried_once_with_safe_memory_settings(monkeypatch, capsys):
    """The first 8-GPU run may be the only one: a rank that runs out of memory kills the job (the others sit in a collective), so
    the parent retries ONCE with every layer re-materialised and says why in the line; a second failure exits non-zero."""
    import bench
    oom = (1, None, ["[rank3]: torch.OutOfMemoryError: HIP out of memory. Tried to allocate 2.00 GiB", "ChildFailedError"])
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "8"], [oom, (0, dict(_LINE, n_gpus=8), [])])
    assert rc == 0 and len(calls) == 2 and len(out) == 1
    second = calls[1][0]
    j = second.index("--remat-free-layers")
    assert second[j:j + 4] == bench.SAFE_MEMORY_ARGS
    assert "OutOfMemoryError" in second[second.index("--retry-reason") + 1]
    # the worker copies --retry-reason into config.retry_reason (bench.main, role worker); both attempts failing: no line, rc 1
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "8"], [oom, oom])
    assert rc == 1 and len(calls) == 2 and not out


def test_one_gpu_default_run_carries_the_other_contexts_as_legs(monkeypatch, capsys):
    """N = 1, default workload: after the 9 s measurement the parent runs BASELINE configs[1] (3 s), the metric's second context
    (63 s), configs[3] (the 30 s stage) and configs[4] (one 63 s denoising step) as legs - own processes - and the CPU baseline, and
    merges them into the ONE line where the driver's record keeps them: `config.legs` + flat `config.leg_*` scalars (round 5: the
    driver kept only the NAMES of other top-level keys); a leg that does not fit the time budget is skipped with a stated reason; a
    failing leg costs nothing but itself."""
    import bench
    leg3 = dict(_LINE, value=7500.0, ms_per_step=2340.0, steps=2, warmup=1, config=dict(_LINE["config"], workload="3sec"))
    leg63 = dict(_LINE, value=6400.0, ms_per_step=53000.0, steps=2, warmup=1, config=dict(_LINE["config"], workload="63sec", remat_free_layers=0, remat_keep=[]))
    leg30 = dict(_LINE, value=6700.0, ms_per_step=24400.0, steps=2, warmup=1, config=dict(_LINE["config"], workload="30sec"))
    samp = {"metric": "sampling_denoising_step_seconds", "value": 21.8, "unit": "s/step (cond+uncond)", "latent_frames_per_s": 11.6,
            "projected_50_step_video_s": 1090.0, "peak_mem_GiB": 101.0,
            "config": {"workload": "sampling 63sec", "layers": 42, "timed_steps": 1, "tokens": 351168, "mini_batches": 21948, "scan_impl": "mfma"}}
    main_line = dict(_LINE, fsdp1={"impl": "flat", "value": 6990.0, "ms_per_step": 7100.0})
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "1", "--steps", "7", "--warmup", "2"],
                                  [(0, main_line, []), (0, leg3, []), (0, leg63, []), (0, leg30, []), (0, samp, []), (0, {"cpu_baseline": {"value": 1.2}}, [])])
    assert rc == 0 and len(out) == 1 and len(calls) == 6
    line = out[0]
    legs = line["config"]["legs"]
    assert list(legs) == ["ctx3s", "ctx63s", "ctx30s", "sample63s"]
    assert line["value"] == 7000.0 and legs["ctx3s"]["value"] == 7500.0 and legs["ctx63s"]["ms_per_step"] == 53000.0
    assert legs["ctx63s"]["ttt_mlp_bwd_ms"] == 10.9 and legs["ctx63s"]["attn_bwd_ms"] == 13.0 and legs["ctx3s"]["steps"] == 2
    assert legs["ctx30s"]["value"] == 6700.0 and legs["sample63s"]["value"] == 21.8 and legs["sample63s"]["latent_frames_per_s"] == 11.6
    assert legs["sample63s"]["valid"] is True and legs["sample63s"]["mini_batches"] == 21948
    c = line["config"]                                     # the flat scalars a parser that drops nested objects still keeps
    assert c["leg_ctx3s_value"] == 7500.0 and c["leg_ctx63s_value"] == 6400.0 and c["leg_ctx30s_value"] == 6700.0 and c["leg_sample63s_value"] == 21.8
    assert c["leg_ctx63s_ms_per_step"] == 53000.0 and c["leg_sample63s_projected_50_step_video_s"] == 1090.0
    assert c["fsdp1"]["value"] == 6990.0 and c["fsdp1_value"] == 6990.0
    assert line["ctx3s"] == legs["ctx3s"] and line["ctx63s"] == legs["ctx63s"]           # (top level too, as in rounds 4 / 5)
    assert line["cpu_baseline"] == {"value": 1.2}
    assert calls[0][0][:2] == [sys.executable, os.path.abspath(bench.__file__)] and calls[0][0][-2:] == ["--role", "worker"]
    c3, c63, c30, cs = calls[1][0], calls[2][0], calls[3][0], calls[4][0]
    assert c3[c3.index("--video-length") + 1] == "3sec" and c63[c63.index("--video-length") + 1] == "63sec" and c30[c30.index("--video-length") + 1] == "30sec"
    # the 63 s leg keeps the attention outputs of every layer, parked in host memory (round 6)
    assert c63[c63.index("--remat-keep") + 1] == "attn" and "--offload-park-kept" in c63 and "--remat-keep-layers" not in c63 and c63[c63.index("--remat-free-layers") + 1] == "0"
    assert "--no-fsdp1-compare" in c3 and c3[-2:] == ["--role", "worker"] and "--remat-free-layers" not in c30
    assert cs[1].endswith(os.path.join("tools", "sample_bench.py")) and cs[cs.index("--video-length") + 1] == "63sec" and cs[cs.index("--steps") + 1] == "1"
    # a subset of the legs
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "1", "--legs", "ctx3s,sample63s", "--no-cpu-baseline"], [(0, _LINE, []), (0, leg3, []), (0, samp, [])])
    assert rc == 0 and len(calls) == 3 and list(out[0]["config"]["legs"]) == ["ctx3s", "sample63s"]
    # no time left: every leg skipped with a reason, the main line and the CPU baseline still there
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "1", "--time-budget", "100"], [(0, _LINE, []), (0, {"cpu_baseline": {"value": 1.2}}, [])])
    assert rc == 0 and len(calls) == 2 and all("budget" in out[0]["config"]["legs"][n]["skipped"] for n in bench.LEG_ORDER)
    assert "budget" in out[0]["config"]["leg_ctx30s_skipped"]
    # a leg that dies: its entry says so
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "1", "--no-cpu-baseline", "--legs", "ctx3s,ctx63s"], [(0, _LINE, []), (1, None, ["HIP error"]), (0, leg63, [])])
    assert rc == 0 and "error" in out[0]["config"]["legs"]["ctx3s"] and out[0]["config"]["legs"]["ctx63s"]["value"] == 6400.0
    assert "HIP error" in out[0]["config"]["leg_ctx3s_error"]
    # the parked 63 s attempt dies: once more with the attention outputs of ten layers on the device, and the entry says why
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "1", "--no-cpu-baseline", "--legs", "ctx63s"], [(0, _LINE, []), (1, None, ["HIP out of memory"]), (0, leg63, [])])
    fb = calls[2][0]
    assert rc == 0 and len(calls) == 3 and fb[fb.index("--remat-keep-layers") + 1] == "10" and "--offload-park-kept" not in fb
    assert out[0]["config"]["legs"]["ctx63s"]["value"] == 6400.0 and "out of memory" in out[0]["config"]["legs"]["ctx63s"]["fallback_after"]
    # a box without 160 GiB of host memory to spare: the parked attempt is not made at all
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "1", "--no-cpu-baseline", "--legs", "ctx63s"], [(0, _LINE, []), (0, leg63, [])], host_gib=62.0)
    only = calls[1][0]
    assert rc == 0 and len(calls) == 2 and "--offload-park-kept" not in only and only[only.index("--remat-keep-layers") + 1] == "10"
    assert "62 GiB" in out[0]["config"]["legs"]["ctx63s"]["fallback_after"]
    # another workload than the metric's: no legs
    calls, out, rc = _orchestrate(monkeypatch, capsys, ["--gpus", "1", "--video-length", "3sec", "--no-cpu-baseline"], [(0, _LINE, [])])
    assert len(calls) == 1 and "ctx3s" not in out[0] and "legs" not in out[0]["config"]


def test_run_child_kills_the_whole_process_group_on_timeout(tmp_path):
    """A child that exceeds its time (for --gpus N > 1: the torch.distributed.run launcher) is killed WITH its descendants - rank
    processes left behind would hold the GPUs under the retry and the legs."""
    import time
    import bench
    pidfile = tmp_path / "grandchild.pid"
    prog = ("import subprocess, sys, time; p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)']); "
            f"open({str(pidfile)!r}, 'w').write(str(p.pid)); time.sleep(600)")
    rc, line, tail = bench.run_child([sys.executable, "-c", prog], timeout=3.0)
    assert rc == -9 and line is None and "process group" in tail[-1]
    gpid = int(pidfile.read_text())
    for _ in range(50):
        try:
            os.kill(gpid, 0)
        except ProcessLookupError:
            break
        # a zombie still answers kill(0): look at its state
        try:
            if open(f"/proc/{gpid}/stat").read().split()[2] == "Z":
                break
        except FileNotFoundError:
            break
        time.sleep(0.1)
    else:
        os.kill(gpid, 9)
        raise AssertionError("the grandchild survived the time-out")


def test_clock_sampler_without_a_gpu_reports_an_error_not_an_exception():
    """bench.ClockSampler is best effort: on a box where amdsmi cannot initialise (no GPU here) the summary says so and start / stop
    are no-ops; with a fake library it averages what it read."""
    import bench
    cs = bench.ClockSampler(0)
    cs.start(); cs.stop()
    s = cs.summary()
    assert "error" in s or "clock_mhz_avg" in s

    class FakeSmi:
        class AmdSmiClkType:
            GFX = 0

        @staticmethod
        def amdsmi_get_clock_info(h, t):
            return {"clk": 2100}

        @staticmethod
        def amdsmi_get_power_info(h):
            return {"current_socket_power": 0xFFFF, "socket_power": 900, "average_socket_power": "N/A"}
    cs._h, cs._smi, cs.period_s = object(), FakeSmi, 0.01
    cs.start()
    import time
    time.sleep(0.1)
    cs.stop()
    s = cs.summary()
    assert s["clock_mhz_avg"] == 2100 and s["power_w_avg"] == 900 and s["samples"] >= 2


def test_under_a_launcher_the_process_is_a_worker(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "run_child", lambda *a, **k: (_ for _ in ()).throw(AssertionError("a worker must not start children")))
    monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    if not __import__("torch").cuda.is_available():
        with pytest.raises(AssertionError, match="needs a GPU"):
            bench.main()


def test_rccl_summary_reads_the_ranks_log_files(tmp_path):
    import bench
    (tmp_path / "rccl.host.1.log").write_text(
        "host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId 1000 - Init START\n"
        "host:1:1 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7\n"
        "host:1:1 [0] NCCL INFO Channel 01/16 :    0   2   4   6   1   3   5   7\n"
        "host:1:1 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 2/-1/-1->0->-1\n")
    r = bench.rccl_summary(str(tmp_path))
    assert r["nranks"] == 8 and r["channels"] == 2 and r["rank_logs"] == 1 and len(r["rings_head"]) == 2 and r["trees_head"]
    assert bench.rccl_summary(str(tmp_path / "nothing")) == {}


def test_sweep_launch_summary_reproduces_from_the_committed_trace():
    """tools/sweep_launches.py on profiles/r4y_ttt_bwd_launches.csv.gz (the sweep / tail / recompute / fill dispatches of round 4's
    rocprofv3 trace of `python bench.py`): the numbers DESIGN.md section 5 quotes - the sweep is unimodal on the replica path and
    bimodal on the sharded path, nothing foreign runs beside it, and 'tail dispatched first' is the slow outcome of the race."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sweep_launches", os.path.join(ROOT, "tools", "sweep_launches.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    res = m.summarize(os.path.join(ROOT, "profiles", "r4y_ttt_bwd_launches.csv.gz"))
    rep, sh = res["phases"]["replica"], res["phases"]["sharded (fsdp1)"]
    assert (rep["launches"], sh["launches"]) == (7392, 3696)
    assert abs(rep["p50"] - 915.3) < 0.5 and rep["p90"] - rep["p10"] < 50            # one mode
    assert abs(sh["p50"] - 1067.1) < 0.5 and sh["p90"] - sh["p10"] > 200             # two
    for p in (rep, sh):                                                              # only the flag memset's tail end overlaps a sweep
        assert set(p["foreign"]) <= {"fill"} and sum(t for _, t in p["foreign"].values()) < 100.0
    first = lambda p, order: sum(v["n"] for (speed, o), v in p["race"].items() if o == order)
    assert first(rep, "tail first") / (first(rep, "tail first") + first(rep, "sweep first")) < 0.04
    assert first(sh, "tail first") / (first(sh, "tail first") + first(sh, "sweep first")) > 0.55
    assert sh["race"][("slow", "tail first")]["n"] > 1500 and sh["race"][("slow", "tail first")]["sweep_us"] > 1100
    assert rep["race"][("fast", "sweep first")]["sweep_us"] < 925
