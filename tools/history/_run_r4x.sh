#!/bin/bash
# Round 4, call X: the final tree - full -m gpu suite, smoke(), the driver's default bench line (with cpu_baseline and the fsdp1
# point), then the SAME command under rocprofv3 --kernel-trace --stats: kernel statistics for profiles/ and the per-launch view of
# the sweep in the replica phase against the fsdp1 phase (tools/sweep_launches.py).
cd /root/repo; mkdir -p gpurun_out/r4x; O=$GRAFT_REPO_ROOT/gpurun_out/r4x
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['steps'], d['config']['remat_free_layers'], d['peak_mem_gib'], 'bwd', r['avg_launch_ms'], r['frac'], {k: round(v['avg_ms'],2) for k,v in r['other'].items()}, 'fsdp1', d.get('fsdp1'), 'cpu', d.get('cpu_baseline'))"
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python /root/repo/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench_rocprof.err; echo "rocprof rc=$?"
cd /root/repo
S=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
[ -n "$S" ] && cp "$S" $O/bench_default_kernel_stats.csv
grep -h "^{" $O/bench_rocprof.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('under rocprof', d['value'], d['ms_per_step'], 'bwd', r['avg_launch_ms'])"
[ -n "$T" ] && python tools/sweep_launches.py "$T" > $O/sweep_launches.txt 2>&1; cat $O/sweep_launches.txt | cut -c1-220
[ -n "$T" ] && python tools/_fmt_overlap.py "$T" > $O/overlap_by_queue.txt 2>&1
[ -n "$T" ] && ls -la "$T" && gzip -c "$T" > $O/kernel_trace.csv.gz && ls -la $O/kernel_trace.csv.gz
