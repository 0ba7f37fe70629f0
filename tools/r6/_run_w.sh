#!/bin/bash
# round 6, call W: one TTT layer direction at the 5B / 9 s geometry - parts x (pair scan | one-workgroup scan)
cd /root/repo; mkdir -p gpurun_out/r6w; O=gpurun_out/r6w
timeout 600 python tools/ttt_layer_bench.py --parts 0,2,3,4,6,8 --rounds 3 > $O/layer_pair.json 2>$O/layer_pair.err; tail -1 $O/layer_pair.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v['median_ms'] for k,v in d['by_parts'].items()})"
timeout 600 python tools/ttt_layer_bench.py --parts 0,4,6 --rounds 3 --debug-option scan_pair=0 > $O/layer_single.json 2>$O/layer_single.err; tail -1 $O/layer_single.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v['median_ms'] for k,v in d['by_parts'].items()})"
