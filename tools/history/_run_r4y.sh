#!/bin/bash
# Round 4, call Y: `python bench.py` under rocprofv3 --kernel-trace --stats (csv): kernel statistics of the final tree and the
# per-launch view of the sweep, replica phase against fsdp1 phase (call X ran the same without --output-format csv).
cd /root/repo; mkdir -p gpurun_out/r4y; O=$GRAFT_REPO_ROOT/gpurun_out/r4y
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python /root/repo/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench_rocprof.err; echo "rocprof rc=$?"
cd /root/repo
S=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); echo "stats=$S trace=$T"
[ -n "$S" ] && cp "$S" $O/bench_default_kernel_stats.csv
grep -h "^{" $O/bench_rocprof.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('under rocprof', d['value'], d['ms_per_step'], 'bwd', r['avg_launch_ms'], 'fsdp1', d.get('fsdp1'))"
if [ -n "$T" ]; then
  python tools/sweep_launches.py "$T" > $O/sweep_launches.txt 2>&1; cut -c1-220 $O/sweep_launches.txt
  python tools/_fmt_overlap.py "$T" > $O/overlap_by_queue.txt 2>&1
  ls -la "$T"; gzip -c "$T" > $O/kernel_trace.csv.gz; ls -la $O/kernel_trace.csv.gz
fi
