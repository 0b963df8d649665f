#!/bin/bash
# round-2 batch 17: scheduler settings for the closest-hit (dual-box) traversal on the HBM-bound soups
set -o pipefail
O=gpurun_out/r2r; mkdir -p $O
for w in soup_1m_ploc_closest_720p_16spp soup_10m_ploc_closest_720p_16spp; do
for k in 4 6 7; do for b in 16 32 64; do for sh in 16 24; do
  SRT_KEEP_EIGHTHS=$k SRT_NODE_BURST=$b SRT_SHADE_MIN=$sh timeout -k 10 300 python bench.py --workload $w --steps 3 --no-cpu-baseline --no-pmc > $O/${w}_${k}_${b}_$sh.json 2>/dev/null
  echo "$w keep $k burst $b shade $sh: $(python -c "import json;d=json.load(open('$O/${w}_${k}_${b}_$sh.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done; done; done; done
for k in 4 6; do for b in 32 64; do
echo "headline scene, tree modes, keep $k burst $b:"; SRT_KEEP_EIGHTHS=$k SRT_NODE_BURST=$b timeout -k 10 300 python tools/tree_modes.py 2>/dev/null | tail -4
done; done
