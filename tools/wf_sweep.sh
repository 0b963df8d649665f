#!/bin/bash
# One-at-a-time sweep of the path-pool kernel's scheduling tunables on the headline frame (1000 spp), Msamples/s.
# usage (GPU box): bash tools/wf_sweep.sh > gpurun_out/wf_sweep.txt
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --no-pmc --no-cpu-baseline --no-hbm-point --steps 3 --spp ${SPP:-1000} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"; }
run X=default
for v in 8 12 24 32; do run SRT_WF_SWAP_MIN=$v; done
for v in 6 8 16 20; do run SRT_PRIM_MIN=$v; done
for v in 32 128; do run SRT_NODE_BURST=$v; done
for v in 3 5 6; do run SRT_KEEP_EIGHTHS=$v; done
for v in 16 24 48; do run SRT_FUSE_MIN=$v; done
for v in 1024; do run SRT_WF_POOL=$v; done
for v in 2 8; do run SRT_PRIM_AGAIN_MIN=$v; done
run X=default
