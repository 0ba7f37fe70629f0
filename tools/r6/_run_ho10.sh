#!/bin/bash
# round 6, call HO10: which engine does the runtime pick for the copies out / in?  (AMD_LOG_LEVEL=4, copy mask only; 6-layer debug model)
cd /root/repo; mkdir -p gpurun_out/r6ho10; O=gpurun_out/r6ho10
AMD_LOG_LEVEL=4 AMD_LOG_MASK=256 timeout 600 python bench.py --role worker --gpus 1 --layers 6 --steps 2 --warmup 1 --no-fsdp1-compare --offload-trace --offload-gib-per-layer 2 --remat-free-layers 6 > $O/bench.json 2> $O/bench.err
grep -h "^{" $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), d['config']['host_offload'])"
grep -c "HSA Copy" $O/bench.err
grep "HSA Copy" $O/bench.err | sed -E 's/dst=0x[0-9a-f]+, src=0x[0-9a-f]+, //; s/wait_event=0x[0-9a-f]+, completion_signal=0x[0-9a-f]+//; s/^.*HSA Copy/HSA Copy/' | awk '{ $NF=""; print }' | sed -E 's/size=([0-9]{1,7}),/size=small,/' | sort | uniq -c | sort -rn | head -30 > $O/copy_engines.txt
cat $O/copy_engines.txt
grep -i "sdma\|falling to Blit\|Max SDMA" $O/bench.err | sort | uniq -c | sort -rn | head -10
grep "HSA Copy" $O/bench.err | grep -v "size=[0-9]\{1,7\}," | head -12
mv $O/bench.err $O/bench_full.err; grep -v "HSA Copy" $O/bench_full.err | tail -50 > $O/bench.err; rm $O/bench_full.err
