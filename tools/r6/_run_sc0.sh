#!/bin/bash
# round 6, call SC0: role B reads the records with sc0 loads on a common XCD - bit identity (stale L1 lines would break it), time, fabric writes
cd /root/repo; mkdir -p gpurun_out/r6sc0; O=$GRAFT_REPO_ROOT/gpurun_out/r6sc0
timeout 900 python - > $O/tests_sc0.log 2>&1 <<'PY'
import sys, pytest
sys.path.insert(0, "ttt-video-dit_amd")
import test_time_training as e
e.load_library(); e.debug_option("scan_b_sc0", 1)
sys.exit(pytest.main(["tests/test_scan_pair_gpu.py", "tests/test_parity_r5_gpu.py", "-x", "-q", "-m", "gpu"]))
PY
echo "tests rc=$?"; tail -3 $O/tests_sc0.log
for rep in 1 2 3; do timeout 300 python tools/op_bench.py --nc 804 --iters 12 --fwd-only --ab scan_b_sc0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ab'])" | tee -a $O/ab_scan_b_sc0.txt; done
export TMPDIR=/tmp; cd /tmp
for v in 0 1; do
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "mlp_scan" --output-format csv -d /tmp/pmc_w$v -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 --fwd-only --ab-fixed scan_b_sc0=$v > /dev/null 2>&1
  f=$(find /tmp/pmc_w$v -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/fwd_pmc_WRITE_SIZE_sc0_$v.csv && python -c "
import csv; r=[float(x['Counter_Value']) for x in csv.DictReader(open('$f')) if x['Counter_Name']=='WRITE_SIZE']; print('scan_b_sc0=$v WRITE_SIZE GB per scan', sum(r)/len(r)*1024/1e9)"
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "mlp_scan" --output-format csv -d /tmp/pmc_f$v -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 --fwd-only --ab-fixed scan_b_sc0=$v > /dev/null 2>&1
  f=$(find /tmp/pmc_f$v -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python -c "
import csv; r=[float(x['Counter_Value']) for x in csv.DictReader(open('$f')) if x['Counter_Name']=='FETCH_SIZE']; print('scan_b_sc0=$v FETCH_SIZE x2 GB per scan', 2*sum(r)/len(r)*1024/1e9)"
done
