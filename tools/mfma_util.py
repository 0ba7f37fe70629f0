#!/usr/bin/env python
"""rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES` CSV -> MFMA utilisation per kernel.

    python tools/mfma_util.py profiles/r3j_op_nc804_pmc_sq.csv [...]

`SQ_VALU_MFMA_BUSY_CYCLES` counts shader cycles in which a SIMD's MFMA pipe is busy, summed over the chip (32 per
v_mfma_f32_32x32x16_bf16: MI355X_MICROARCH.md, cycle-constant table).  Utilisation = that sum / (dispatch duration in shader
cycles at 2.4 GHz x SIMDs): of the whole chip (1 024 SIMDs) and of the SIMDs the launch occupies (4 per workgroup's CU, from the
grid: one workgroup per CU for these kernels unless the grid exceeds 256).  `SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES` = share of a
resident wave's time spent issue-stalled (both in quad-cycles)."""
import collections
import csv
import sys

CLK = 2.4e9
for path in sys.argv[1:]:
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = disp[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d["wgs"] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in disp.values():
        a = agg[d["name"]]
        a["n"] += 1
        a["dur"] += d["dur"]
        a["cap_chip"] += d["dur"] * CLK * 1024
        a["cap_occ"] += d["dur"] * CLK * 4 * min(d["wgs"], 256)
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
            a[k] += d.get(k, 0.0)
        a["wgs"] = d["wgs"]
    print(path)
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
        if a["SQ_WAVE_CYCLES"] == 0:
            continue
        print(f"  {name[-58:]:58s} launches {int(a['n']):4d}  avg {1e3 * a['dur'] / a['n']:7.3f} ms  workgroups {int(a['wgs']):5d}  "
              f"MFMA busy: {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['cap_chip']:5.2f} % of the chip, "
              f"{100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['cap_occ']:5.2f} % of the occupied SIMDs   "
              f"issue-stalled {100 * a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']:4.1f} % of wave time")
