"""Where do a kernel's waves spend their cycles?  Reads a rocprofv3 --pmc pass with
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
(tools/_run_r3p.sh) and prints per kernel: the parked / issue-stalled / issuing shares of the wave cycles (disjoint, guide
MI355X_MICROARCH.md "rocprofv3 PMC slots"), the LDS-issue-stall sub-share, and the LDS array's busy share of the occupied CUs'
cycles with the part of it that is bank-conflict replays.

    python tools/wait_lds.py profiles/r3p_op_nc804_pmc_wait_lds.csv [--clock-ghz 2.4]
"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n, dur, wgs = collections.Counter(), collections.defaultdict(float), {}
    for r in csv.DictReader(open(a.csv)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        wgs[k] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"{'kernel':58s} {'n':>3s} {'us':>8s} {'parked':>7s} {'stalled':>7s} {'(LDS)':>6s} {'issuing':>7s} {'LDS busy':>8s} {'conflicts':>9s}")
    for k, c in agg.items():
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        cu_cycles = dur[k] * a.clock_ghz * min(256, wgs[k])           # ns x GHz = cycles, x CUs the launch can occupy
        print(f"{k[-58:]:58s} {n[k]:3d} {dur[k] / n[k] / 1e3:8.1f} {c['SQ_WAIT_ANY'] / wc:7.1%} {c['SQ_WAIT_INST_ANY'] / wc:7.1%} "
              f"{c['SQ_WAIT_INST_LDS'] / wc:6.1%} {c['SQ_ACTIVE_INST_ANY'] / wc:7.1%} {c['SQ_LDS_IDX_ACTIVE'] / cu_cycles:8.1%} "
              f"{c['SQ_LDS_BANK_CONFLICT'] / cu_cycles:9.1%}")


if __name__ == "__main__":
    main()
