#!/bin/bash
# round 6, call I: what a remat-free layer saves (tools/saved_tensor_audit.py), attention priority pair final A/B, attention tests
cd /root/repo; mkdir -p gpurun_out/r6i; O=$GRAFT_REPO_ROOT/gpurun_out/r6i
timeout 600 python tools/saved_tensor_audit.py > $O/saved_tensor_audit.json 2> $O/audit.err; echo "audit rc=$?"; tail -3 $O/audit.err
python - <<PY
import json
d=json.load(open("$O/saved_tensor_audit.json"))
print({k: v for k, v in d.items() if k != "storages"})
for e in d["storages"]: print(e["GiB"], e["units_LD_bf16"], e["n_saves"], e["views"][:2], e["saved_by"][:3])
PY
timeout 600 python tools/attn_bench.py --no-sdpa --iters 6 --variants 0,1,0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['variants'].items(): print('attn bwd prio',k, v['bwd_ms'], v['bwd_ms_rounds'], v['dq_dk_dv_equal_to_first'])"
timeout 900 python -m pytest tests/test_attention_gpu.py -x -q -m gpu 2>&1 | tail -2
