#!/bin/bash
# round-2 batch 18: parity suite and the soup workloads with the mode-dependent scheduler defaults
set -o pipefail
O=gpurun_out/r2s; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "passed\|failed\|Error\|^E  " $O/pytest.txt | tail -8
for w in soup_1m_720p_16spp soup_4m_720p_16spp soup_10m_720p_16spp soup_1m_ploc_closest_720p_16spp soup_4m_ploc_closest_720p_16spp soup_10m_ploc_closest_720p_16spp; do
  timeout -k 10 900 python bench.py --workload $w --steps 3 --no-cpu-baseline --save-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(python -c "import json;d=json.load(open('$O/bench_$w.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], 'alg', r.get('algorithmic_GBps'), r.get('frac'), 'hbm', r.get('hbm_measured_GBps'), r.get('hbm_measured_frac'))" 2>&1)"
done
timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/head.json 2>/dev/null; python -c "import json;d=json.load(open('$O/head.json'));print('headline', d['value'], d['roofline']['kernel_ms_avg'])"
cp profiles/pmc_*.json $O/ 2>/dev/null
# mini-sweep around the new scheduler defaults (headline frame)
for v in "" unroll6 unroll8 prim3; do
  if [ -n "$v" ]; then export SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_$v.so; else unset SRT_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/knob_$v.json 2>/dev/null
  echo "variant '$v': $(python -c "import json;d=json.load(open('$O/knob_$v.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done
unset SRT_HIP_LIB
for e in "SRT_SHADE_MIN=32" "SRT_HIT_MIN=32" "SRT_HIT_MIN=32 SRT_SHADE_MIN=32" "SRT_KEEP_EIGHTHS=3 SRT_NODE_BURST=128" "SRT_PRIM_MIN=16" "SRT_FUSE_MIN=24"; do
  env $e timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/env.json 2>/dev/null
  echo "$e: $(python -c "import json;d=json.load(open('$O/env.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done
