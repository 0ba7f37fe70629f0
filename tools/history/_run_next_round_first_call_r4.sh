#!/bin/bash
# Prepared at the end of round 4 (GPU budget spent): the measurements that decide round 4's open items, for the FIRST gpurun call
# of the next round (~17 GPU-minutes: the suite 3.5, five bench runs of 2.7; one box, back to back).
#   1. the full GPU suite on the final tree (incl. test_tail_gate_delay_same_bits, which has not run on a device yet)
#   2. THE open question of the sharded path (profiles/r4y_sweep_launches.txt): on the fsdp1 point two sweep
#      launches in three run 1.12 instead of 0.92 ms, paired with a slower tail kernel beside them, with no third kernel involved.
#      The trace's offline reading (profiles/r4zc_flags_memset_early_op_level.txt): "tail dispatched first" is the slow outcome, and
#      the flag memset between recompute and sweep makes it a coin toss.  Two remedies, both bit-identical; the first became the
#      default at the end of round 4 WITHOUT an in-step measurement (op level only) - this call is that measurement:
#        TTT_FLAGS_MEMSET_EARLY=1   the memset moved behind the previous sweep: the sweep follows its recompute kernel-to-kernel
#        TTT_TAIL_DELAY_US=25       a gate kernel in front of the tail
#        TTT_TAIL_GATE_RESIDENT=1   the gate waits until the next sweep's workgroups have all started (never run on a device yet)
#      (bench.py hands both to ext.debug_option); fsdp1.ttt_mlp_bwd_ms: 12.3 -> 10.9 ?  Adopt the winner as the default if the fsdp1
#      point comes within 1 % of `value` AND the replica line does not move (it should gain too: 3 % of its launches are slow).  If it does not help: rocprofv3 --kernel-trace of that run, then
#      tools/sweep_launches.py - compare which CUs / XCDs the tail's workgroups get in fast and slow launches
#      (--kernel-trace has no CU ids: add `s_getreg HW_ID` stamps of workgroup 0..7 to the sweep's dbg buffer).
#   3. (if an 8-GPU node is not available again) the same trace with FLAT_FSDP_DEFER_REDUCE=1 at N = 1 with collectives, to
#      confirm that only the timing of the reduce decides the regime.
cd /root/repo; mkdir -p gpurun_out/r5a; O=$GRAFT_REPO_ROOT/gpurun_out/r5a
export TMPDIR=/tmp
TTT_TEST_VARIANTS=1 timeout 600 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_suite.log
for cfg in "TTT_FLAGS_MEMSET_EARLY=0" "TTT_FLAGS_MEMSET_EARLY=1" "TTT_FLAGS_MEMSET_EARLY=0 TTT_TAIL_DELAY_US=25" "TTT_FLAGS_MEMSET_EARLY=1 TTT_TAIL_DELAY_US=25" "TTT_TAIL_GATE_RESIDENT=1"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$cfg rc=$?"
  grep -h "^{" $O/bench_$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'ttt bwd', round(r['avg_launch_ms'],3), 'fsdp1', d.get('fsdp1'))"
done
