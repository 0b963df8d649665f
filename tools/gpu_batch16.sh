#!/bin/bash
# round-2 batch 16: node-burst length sweep (keep fraction x burst budget x restart threshold)
set -o pipefail
O=gpurun_out/r2q; mkdir -p $O
SWEEP_CHUNKS=0 SWEEP_SHADE=16,24 SWEEP_PRIM=12 SWEEP_BURST=32,64,128 SWEEP_HIT=24 SWEEP_FUSE=32 SWEEP_AGAIN=4 SWEEP_KEEP=0,2,3,4,5,6 timeout -k 10 900 python tools/sweep.py masterchief 2000 > $O/sweep1.txt 2>&1; sort -k18 -n -r $O/sweep1.txt | head -10; sort -k18 -n $O/sweep1.txt | head -3
SWEEP_CHUNKS=0 SWEEP_SHADE=24,32 SWEEP_PRIM=8,12,20 SWEEP_BURST=64 SWEEP_HIT=24,32,40 SWEEP_FUSE=16,32 SWEEP_AGAIN=4 SWEEP_KEEP=3 timeout -k 10 900 python tools/sweep.py masterchief 2000 > $O/sweep2.txt 2>&1; sort -k18 -n -r $O/sweep2.txt | head -6
