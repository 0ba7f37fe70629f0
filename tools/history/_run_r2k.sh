#!/bin/bash
# (historical: the --overlap modes / --side-wgs option of tools/op_bench.py that this call exercised were removed with the schedule they tested; results under profiles/)
# round 2, call K: does the group recompute really run UNDER the sweep?  Kernel trace (begin / end timestamps per dispatch) of
# tools/op_bench.py at the 9 s scan length with the side-stream schedule, and the sweep's stage cycle stamps with / without it
mkdir -p gpurun_out/r2k
O=$GRAFT_REPO_ROOT/gpurun_out/r2k
for ov in 0 1; do
  timeout 300 python tools/op_bench.py --nc 804 --overlap $ov --iters 4 --phases 2>/dev/null | python tools/_fmt_phases.py "overlap $ov" | tee -a $O/phases.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --overlap 1 --iters 2 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
echo "trace: $f"; head -1 "$f"
python - "$f" > $O/overlap_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ks.sort()
# last backward: take the last 40 dispatches of the three kernels
sel = [k for k in ks if "mlp_scan_kernel" in k[2] or "mlp_bwd_cluster" in k[2] or "mlp_bwd_tail" in k[2]][-36:]
t0 = sel[0][0]
for s, e, n, q, st in sel:
    short = "A recompute" if "mlp_scan_kernel" in n else ("B sweep" if "cluster" in n else "C tail")
    print(f"{(s - t0) / 1e3:10.1f} us  -> {(e - t0) / 1e3:10.1f} us   dur {(e - s) / 1e3:8.1f} us   queue {q} stream {st}  {short}")
PY
cat $O/overlap_trace.txt | head -40
