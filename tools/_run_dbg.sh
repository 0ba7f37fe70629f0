#!/bin/bash
# DEBUG helper for gpurun: op-level timing of the scan and attention kernels
mkdir -p gpurun_out/dbg
timeout 120 python tools/op_bench.py --phases --iters 5 > gpurun_out/dbg/op.json 2>&1
tail -1 gpurun_out/dbg/op.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bwd ms', round(d['bwd']['avg_ms'],3), 'fwd ms', round(d['fwd']['avg_ms'],3), [int(x) for x in d['phase_cycles_per_step'][16:26]])"
timeout 200 python tools/attn_bench.py --no-sdpa --iters 7 2>/dev/null | tail -1
timeout 200 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -1
