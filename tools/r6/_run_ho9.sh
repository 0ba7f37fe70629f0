#!/bin/bash
# round 6, call HO9: memory-copy + kernel trace of a 9 s step with 8 layers x 2 GiB offloaded: how long does each copy out really take?
cd /root/repo; mkdir -p gpurun_out/r6ho9; O=gpurun_out/r6ho9
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --memory-copy-trace --output-format csv -d /tmp/ho9 -o t -- python /root/repo/bench.py --role worker --gpus 1 --steps 2 --warmup 1 --no-fsdp1-compare --offload-trace --offload-gib-per-layer 2 --offload-backlog-gib 64 --offload-layers 8 --remat-free-layers 16 > /root/repo/$O/bench.json 2> /root/repo/$O/bench.err
cd /root/repo
f=$(find /tmp/ho9 -name "*memory_copy_trace.csv" | head -1); echo $f; ls -la $(dirname $f)
python tools/copy_trace.py $f | tee $O/copy_trace.txt
head -3 $f > $O/copy_head.csv
grep -h "^{" $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), d['config']['host_offload'])"
# ... and the baseline with 2 / 3 / 4 hardware queues (HO7: 8 queues cost 4.6 %)
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -3 ${1%.json}.err; }
run() { timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare "${@:2}" > $O/bench_$1.json 2> $O/bench_$1.err; show $O/bench_$1.json $1; }
run q4 --remat-free-layers 14
GPU_MAX_HW_QUEUES=2 run q2 --remat-free-layers 14
GPU_MAX_HW_QUEUES=3 run q3 --remat-free-layers 14
run q4b --remat-free-layers 14
