#!/bin/bash
# Every bench.py workload with in-run counters, one JSON line each -> gpurun_out/<dir>/bench_<workload>.json
# usage (GPU box): bash tools/bench_all.sh <dir> [workload ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
W=${@:-masterchief_720p_5000spp iron_720p_5000spp spheres_720p_1024spp spheres_240p_64spp masterchief_1080p_8192spp sphere_field_720p_1024spp soup_1m_720p_16spp soup_4m_720p_16spp soup_10m_720p_16spp soup_1m_ploc_closest_720p_16spp soup_4m_ploc_closest_720p_16spp soup_10m_ploc_closest_720p_16spp}
for w in $W; do
  extra="--no-cpu-baseline --no-hbm-point"; [ $w = masterchief_720p_5000spp ] && extra=""
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 $extra > $O/bench_$w.json 2> $O/bench_$w.err
  rc=$?; echo "[$(date +%T)] rc=$rc $w: $(cut -c1-110 $O/bench_$w.json | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; exit 1; fi
done
