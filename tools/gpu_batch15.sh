#!/bin/bash
# round-2 batch 15: scheduler sweep after the primitive rounds (fuse / again / keep / burst / prim thresholds)
set -o pipefail
O=gpurun_out/r2p; mkdir -p $O
SWEEP_CHUNKS=0 SWEEP_SHADE=16 SWEEP_PRIM=8,12,16 SWEEP_BURST=32,64 SWEEP_HIT=24 SWEEP_FUSE=16,32,48 SWEEP_AGAIN=4 SWEEP_KEEP=1,2,3 timeout -k 10 900 python tools/sweep.py masterchief 2000 > $O/sweep1.txt 2>&1; sort -k18 -n -r $O/sweep1.txt | head -8; sort -k18 -n $O/sweep1.txt | head -3
SWEEP_CHUNKS=0 SWEEP_SHADE=8,16,24 SWEEP_PRIM=12 SWEEP_BURST=32 SWEEP_HIT=16,24,32 SWEEP_FUSE=32 SWEEP_AGAIN=2,4,8 SWEEP_KEEP=2 timeout -k 10 900 python tools/sweep.py masterchief 2000 > $O/sweep2.txt 2>&1; sort -k18 -n -r $O/sweep2.txt | head -6
