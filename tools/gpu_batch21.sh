#!/bin/bash
# round-2 batch 21: scheduler thresholds of the LDS-resident-tree kernel (headline frame, 512 spp), one at a time
set -o pipefail
O=gpurun_out/r2x; mkdir -p $O
export SWEEP_CHUNKS=16
B="SWEEP_SHADE=24 SWEEP_PRIM=12 SWEEP_BURST=64 SWEEP_HIT=24 SWEEP_FUSE=32 SWEEP_AGAIN=4 SWEEP_KEEP=4"
run() { env $B "$@" timeout -k 10 200 python tools/sweep.py masterchief 512 2>&1 | grep Msamples | tee -a $O/sweep.txt; }
run SWEEP_SHADE=12,16,24,32,40
run SWEEP_PRIM=6,8,12,16,20,28
run SWEEP_BURST=8,16,32,64,128
run SWEEP_HIT=12,16,24,32,40
run SWEEP_FUSE=16,24,32,40,48
run SWEEP_KEEP=2,3,4,5,6,7
run SWEEP_AGAIN=2,4,8,16
