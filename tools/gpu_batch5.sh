#!/bin/bash
# round-2 batch 5: exact chunk sums + multiply-high decomposition: parity suite, headline with counters, chunk sweep, configs
set -o pipefail
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "published-image residuals\|passed\|failed\|Error\|^E  " $O/pytest.txt | tail -25
timeout -k 10 600 python bench.py --steps 2 --save-pmc > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_headline.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], 'frac', r.get('frac'), 'issue', r.get('issue_frac'), 'lanes', r.get('lane_utilisation'), 'valu/sample', r.get('valu_wave_instr_per_sample'), 'wait', r.get('wait_frac'), 'traffic', r.get('traffic'))"
timeout -k 10 300 python tools/profile_steps.py masterchief 256 > $O/steps.txt 2>&1; tail -9 $O/steps.txt
SWEEP_CHUNKS=16,32,64,128,157,256,512,1024 SWEEP_SHADE=16 SWEEP_PRIM=12 SWEEP_BURST=32 SWEEP_HIT=24 timeout -k 10 300 python tools/sweep.py masterchief 5000 > $O/sweep_chunks.txt 2>&1; cat $O/sweep_chunks.txt | tail -9
for w in iron_720p_5000spp spheres_720p_1024spp spheres_240p_64spp sphere_field_720p_1024spp masterchief_1080p_8192spp; do
  timeout -k 10 600 python bench.py --workload $w --steps 2 --no-cpu-baseline --save-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(python -c "import json;d=json.load(open('$O/bench_$w.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], r.get('frac'), r.get('lane_utilisation'), r.get('hbm_measured_frac'))" 2>&1)"
done
cp profiles/pmc_*.json $O/ 2>/dev/null
