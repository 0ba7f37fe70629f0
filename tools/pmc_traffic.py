#!/usr/bin/env python
"""rocprofv3 PMC passes -> per-launch HBM traffic summary (profiles/*pmc_traffic*.json, read by bench.py for roofline.traffic).

    python tools/pmc_traffic.py --fetch F.csv --write W.csv --b 1 --nh 48 --nc 804 --calls 4 --out profiles/r2_pmc_traffic_nc804.json

F.csv / W.csv: `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex mlp_ --output-format csv -- python tools/op_bench.py
--nc NC --iters 2` (SEPARATE passes, as MI355X_MICROARCH.md prescribes; `--calls` = forward/backward pairs the command runs:
iters + 2 warm-ups).  Units: the counters are KiB; on gfx950 FETCH_SIZE reads half of what a wide coalesced stream fetches
(MI355X_MICROARCH.md, HBM section) -> fetch bytes = FETCH_SIZE x 2 x 1024, write bytes = WRITE_SIZE x 1024."""
import argparse
import collections
import csv
import json


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            tot[row["Kernel_Name"]] += float(row["Counter_Value"])
            cnt[row["Kernel_Name"]] += 1
    return tot, cnt


def short(name):
    for key, tag in (("mlp_bwd_cluster4_kernel", "sweep_cluster"), ("mlp_recompute8_kernel", "recompute"), ("mlp_bwd_tail4_kernel", "tail"), ("mlp_bwd_tail5_kernel", "tail"),
                     ("mlp_bwd_cluster_kernel", "sweep_cluster"), ("mlp_scan_kernel", "recompute"), ("mlp_bwd_tail_kernel", "tail"),
                     ("mlp_scan8_kernel", "forward_scan"), ("mlp_scan_pair_kernel", "forward_scan")):
        if key in name:
            return tag
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--nh", type=int, default=48)
    ap.add_argument("--nc", type=int, required=True)
    ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    ft, fc = per_kernel(a.fetch, "FETCH_SIZE")
    wt, wc = per_kernel(a.write, "WRITE_SIZE")
    parts = {}
    for name in ft:
        tag = short(name)
        if tag is None:
            continue
        parts[tag] = {"launches_per_call": fc[name] / a.calls, "fetch_bytes_per_launch": ft[name] / fc[name] * 2 * 1024,
                      "write_bytes_per_launch": wt.get(name, 0.0) / max(wc.get(name, 1), 1) * 1024}
    g = 2.0 * 64 * 64 * 256
    bwd = sum(p["launches_per_call"] * (p["fetch_bytes_per_launch"] + p["write_bytes_per_launch"]) for t, p in parts.items() if t != "forward_scan")
    kernels = {"ttt_mlp_bwd_scan[mfma]": {"traffic_bytes_per_backward": bwd, "per_launch": {t: p for t, p in parts.items() if t != "forward_scan"},
                                           "algorithmic_bytes": a.b * a.nh * a.nc * (57.6e3 + 132352 / 16)}}
    if "forward_scan" in parts:
        p = parts["forward_scan"]
        kernels["ttt_mlp_fwd_scan[mfma]"] = {"fetch_bytes": p["fetch_bytes_per_launch"], "write_bytes": p["write_bytes_per_launch"],
                                              "algorithmic_bytes": a.b * a.nh * a.nc * (32896 + 8272)}
    out = {"source": f"{a.fetch}, {a.write} (rocprofv3 --pmc, separate passes; tools/op_bench.py --nc {a.nc})",
           "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section); unit KiB -> bytes x1024; WRITE_SIZE uncorrected",
           "geometry": {"B": a.b, "NH": a.nh, "NC": a.nc}, "kernels": kernels}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
