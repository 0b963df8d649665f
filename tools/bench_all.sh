#!/bin/bash
# Every bench.py workload with in-run counters, one JSON line each -> gpurun_out/<dir>/bench_<workload>.json
# usage (GPU box): [SAVE_PMC=1] bash tools/bench_all.sh <dir> [workload ...]
# SAVE_PMC=1: also refresh the fallback counter records profiles/pmc_<workload>.json (copied to gpurun_out/<dir>/pmc/, the
# only directory that travels back from the box)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
W=${@:-masterchief_720p_5000spp iron_720p_5000spp spheres_720p_1024spp spheres_240p_64spp masterchief_1080p_8192spp sphere_field_720p_1024spp army_720p_1024spp soup_16k_720p_64spp soup_50k_720p_64spp soup_200k_720p_64spp soup_1m_720p_16spp soup_4m_720p_16spp soup_10m_720p_16spp soup_1m_ploc_closest_720p_16spp soup_4m_ploc_closest_720p_16spp soup_10m_ploc_closest_720p_16spp}
for w in $W; do
  extra="--no-cpu-baseline --no-hbm-point"; [ $w = masterchief_720p_5000spp ] && extra=""
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 $extra ${SAVE_PMC:+--save-pmc} > $O/bench_$w.json 2> $O/bench_$w.err
  rc=$?; echo "[$(date +%T)] rc=$rc $w: $(cut -c1-110 $O/bench_$w.json | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out -- stopping"; exit 1; fi
done
if [ -n "$SAVE_PMC" ]; then mkdir -p $O/pmc && cp profiles/pmc_*.json $O/pmc/; fi
