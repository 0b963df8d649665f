#!/bin/bash
# One-at-a-time sweep of the path-pool kernel's scheduling tunables on the headline frame (1000 spp), Msamples/s.
# usage (GPU box): bash tools/wf_sweep.sh > gpurun_out/wf_sweep.txt
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --no-pmc --no-cpu-baseline --no-hbm-point --steps 3 --spp ${SPP:-1000} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"; }
run X=default
for v in 16 24 40 48 56; do run SRT_WF_SWAP_MIN=$v; done
for v in 3; do run SRT_KEEP_EIGHTHS=$v; done
run SRT_WF_SWAP_MIN=40 SRT_KEEP_EIGHTHS=3
run SRT_WF_SWAP_MIN=48 SRT_KEEP_EIGHTHS=3
run SRT_WF_SWAP_MIN=40 SRT_PRIM_MIN=16
run SRT_WF_SWAP_MIN=40 SRT_PRIM_MIN=8
run SRT_WF_SWAP_MIN=40 SRT_NODE_BURST=32
run SRT_WAVEFRONT=0
run X=default
