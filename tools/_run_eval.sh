#!/bin/bash
# gpurun helper: GPU validation + profiles of the evaluation / sampling path (CS = 16 MFMA scan, sampler):
# pytest -m gpu, smoke, rocprofv3 kernel stats of tools/cs16_bench.py (96 scans = the batched guidance pair) and of
# tools/sample_bench.py, SQ counters of the scan kernel; outputs under gpurun_out/eval/
mkdir -p gpurun_out/eval
R=$PWD
timeout 500 python -m pytest tests -m gpu -x -q > gpurun_out/eval/pytest_gpu.log 2>&1
tail -3 gpurun_out/eval/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/eval/smoke.log 2>&1
tail -3 gpurun_out/eval/smoke.log
timeout 100 python tools/cs16_bench.py --batch 2 --phases > gpurun_out/eval/cs16_bench_b2.txt 2>&1
tail -4 gpurun_out/eval/cs16_bench_b2.txt
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof16 -o cs16 -- python $R/tools/cs16_bench.py --batch 2 --no-generic --iters 10 > $R/gpurun_out/eval/cs16_prof.log 2>&1
cp $(find /tmp/prof16 -name "*kernel_stats.csv") $R/gpurun_out/eval/cs16_kernel_stats.csv
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "mlp_scan16" --output-format csv -d /tmp/pmc16 -o cs16 -- python $R/tools/cs16_bench.py --batch 2 --no-generic --iters 2 > $R/gpurun_out/eval/cs16_sq.log 2>&1
cp $(find /tmp/pmc16 -name "*counter_collection.csv") $R/gpurun_out/eval/cs16_pmc_sq.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -o sample -- python $R/tools/sample_bench.py --steps 2 2> $R/gpurun_out/eval/sample_prof.err | grep '^{"metric' > $R/gpurun_out/eval/sample_prof.json
cp $(find /tmp/profs -name "*kernel_stats.csv") $R/gpurun_out/eval/sample_kernel_stats.csv
cd $R
head -12 gpurun_out/eval/sample_kernel_stats.csv | cut -c1-150
ls gpurun_out/eval
