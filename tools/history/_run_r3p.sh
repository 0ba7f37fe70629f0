#!/bin/bash
# call P: where do the waves of the dominant kernels wait?  LDS bank conflicts / LDS issue stalls / parked / issue-stalled shares
# (one SQ pass per benchmark program), and the list of gfx950 counters for later passes
mkdir -p gpurun_out/r3p
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
cd /tmp && export TMPDIR=/tmp
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
timeout 150 rocprofv3 --pmc $C --kernel-include-regex "attn_" --output-format csv -d /tmp/pa -- python $GRAFT_REPO_ROOT/tools/attn_bench.py > /tmp/pa.log 2>&1
f=$(find /tmp/pa -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/attn_pmc_wait_lds.csv || tail -5 /tmp/pa.log
timeout 150 rocprofv3 --pmc $C --kernel-include-regex "mlp_" --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pm.log 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_wait_lds.csv || tail -5 /tmp/pm.log
timeout 60 rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > $O/gfx950_counter_names.txt
wc -l $O/*
