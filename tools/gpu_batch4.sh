#!/bin/bash
# round-2 batch 4: two-paths-per-lane kernel: parity suite, A/B against the one-path kernel, threshold sweep
set -o pipefail
O=gpurun_out/r2e; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "published-image residuals\|passed\|failed\|Error\|^E  " $O/pytest.txt | tail -25
for k in 1 2; do
  SRT_KERNEL=$k timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline > $O/bench_kernel$k.json 2> $O/bench_kernel$k.err
  echo "kernel $k rc=$? $(python -c "import json;d=json.load(open('$O/bench_kernel$k.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], 'frac', r.get('frac'), 'issue', r.get('issue_frac'), 'lanes', r.get('lane_utilisation'), 'valu/sample', r.get('valu_wave_instr_per_sample'), 'wait', r.get('wait_frac'), 'traffic', r.get('traffic'))" 2>&1)"
done
timeout -k 10 300 python tools/profile_steps.py masterchief 256 > $O/steps_kernel2.txt 2>&1; tail -9 $O/steps_kernel2.txt
SWEEP_KERNEL=2 SWEEP_CHUNKS=0 SWEEP_SHADE=16,32,48 SWEEP_PRIM=8,12,20 SWEEP_BURST=32 SWEEP_HIT=24,40,56 SWEEP_SWAP=1,8,16 timeout -k 10 600 python tools/sweep.py masterchief 2000 > $O/sweep_k2.txt 2>&1; sort -k17 -n -r $O/sweep_k2.txt | head -8; sort -k17 -n $O/sweep_k2.txt | head -2
for w in iron_720p_5000spp spheres_720p_1024spp spheres_240p_64spp soup_1m_720p_16spp soup_10m_720p_16spp soup_10m_ploc_closest_720p_16spp; do
  timeout -k 10 600 python bench.py --workload $w --steps 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(python -c "import json;d=json.load(open('$O/bench_$w.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], r.get('frac'), r.get('lane_utilisation'), r.get('hbm_measured_frac'))" 2>&1)"
done
