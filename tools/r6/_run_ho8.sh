#!/bin/bash
# round 6, call HO8: which kernels of the step slow the copies out (tools/pcie_probe3.py), with 4 and 8 hardware queues
cd /root/repo; mkdir -p gpurun_out/r6ho8; O=gpurun_out/r6ho8
timeout 300 python tools/pcie_probe3.py > $O/probe3_q4.json 2> $O/probe3_q4.err; cat $O/probe3_q4.json; tail -3 $O/probe3_q4.err
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/pcie_probe3.py > $O/probe3_q8.json 2> $O/probe3_q8.err; cat $O/probe3_q8.json; tail -3 $O/probe3_q8.err
