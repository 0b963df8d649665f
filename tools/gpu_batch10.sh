#!/bin/bash
# round-2 batch 10: chunk count x exact-sum path on the whole frame and on an 8-rank share (same box)
set -o pipefail
O=gpurun_out/r2k; mkdir -p $O
for c in 64 157 256; do for mb in 0 8192; do
  SRT_CHUNK_SCRATCH_MB=$mb timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc --spp-chunks $c > $O/head_${c}_$mb.json 2>/dev/null
  echo "headline chunks $c budget $mb MiB: $(python -c "import json;d=json.load(open('$O/head_${c}_$mb.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done; done
for c in 64 157 256; do for mb in 0 8192; do
  echo "8-rank share, chunks $c budget $mb:"; SRT_CHUNK_SCRATCH_MB=$mb CHUNKS=$c RANKS=1,8 timeout -k 10 300 python tools/partition_balance.py masterchief 5000 2>/dev/null | cut -c1-260
done; done
for c in 64 256; do
  SRT_CHUNK_SCRATCH_MB=0 timeout -k 10 300 python bench.py --workload masterchief_1080p_8192spp --steps 1 --no-cpu-baseline --no-pmc --spp-chunks $c > $O/c5_${c}.json 2>/dev/null
  echo "1080p chunks $c atomic: $(python -c "import json;d=json.load(open('$O/c5_${c}.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done
