#!/bin/bash
# round-2 batch 14: closest-hit traversal over 64-byte dual-box records vs the 32-byte records (before_dual), parity suite
set -o pipefail
O=gpurun_out/r2o; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "passed\|failed\|Error\|^E  " $O/pytest.txt | tail -12
for w in soup_1m_ploc_closest_720p_16spp soup_10m_ploc_closest_720p_16spp; do for v in "" before_dual; do
  if [ -n "$v" ]; then export SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_$v.so; else unset SRT_HIP_LIB; fi
  timeout -k 10 900 python bench.py --workload $w --steps 3 --no-cpu-baseline > $O/${w}_$v.json 2>/dev/null
  echo "$w '$v': $(python -c "import json;d=json.load(open('$O/${w}_$v.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], 'hbm', r.get('hbm_measured_GBps'), r.get('hbm_measured_frac'), 'alg', r.get('algorithmic_GBps'), 'visits/ray', r.get('node_visits_per_ray'))" 2>&1)"
done; done
unset SRT_HIP_LIB
timeout -k 10 300 python tools/tree_modes.py > $O/tree_modes.txt 2>&1; tail -8 $O/tree_modes.txt
timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/head.json 2>/dev/null; python -c "import json;d=json.load(open('$O/head.json'));print('headline', d['value'], d['roofline']['kernel_ms_avg'])"
