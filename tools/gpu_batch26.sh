#!/bin/bash
# round-2 batch 26: scheduler thresholds and node-loop unroll of the LDS-tree kernel at the full 5000 spp
set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
run() { timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/x.json 2>/dev/null; echo "$1: $(python -c "import json;d=json.load(open('$O/x.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)" | tee -a $O/knobs.txt; }
run "default"
for e in "SRT_SHADE_MIN=16" "SRT_SHADE_MIN=20" "SRT_PRIM_MIN=8" "SRT_FUSE_MIN=24" "SRT_HIT_MIN=20" "SRT_SHADE_MIN=16 SRT_PRIM_MIN=8 SRT_FUSE_MIN=24" "SRT_NODE_BURST=32" "SRT_KEEP_EIGHTHS=3"; do
  export $e; run "$e"; for kv in $e; do unset ${kv%%=*}; done
done
for u in 2 8; do export SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_unroll$u.so; run "unroll$u"; unset SRT_HIP_LIB; done
run "default again"
