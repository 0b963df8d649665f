#!/bin/bash
# round-2 final measurement set (after the LDS-resident tree and the flattened shading records): parity suite, the driver's default bench run, rocprofv3 kernel stats, counters of every
# workload, partition balance
set -o pipefail
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "passed\|failed\|Error\|^E  " $O/pytest.txt | tail -12
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python bench.py --save-pmc > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_default.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], 'frac', r.get('frac'), 'issue', r.get('issue_frac'), 'lanes', r.get('lane_utilisation'), 'valu/sample', r.get('valu_wave_instr_per_sample'), 'traffic', r.get('traffic'), d['config']['spp_chunks'])"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err ); echo "rocprof rc=$?"; find $O/kstats -name "*kernel_stats.csv" | head -1 | while read f; do cp $f $O/kernel_stats.csv; head -4 $f; done
timeout -k 10 300 python tools/profile_steps.py masterchief 256 > $O/steps.txt 2>&1; tail -8 $O/steps.txt
for w in iron_720p_5000spp spheres_720p_1024spp spheres_240p_64spp sphere_field_720p_1024spp masterchief_1080p_8192spp soup_1m_720p_16spp soup_4m_720p_16spp soup_10m_720p_16spp soup_1m_ploc_closest_720p_16spp soup_4m_ploc_closest_720p_16spp soup_10m_ploc_closest_720p_16spp; do
  timeout -k 10 900 python bench.py --workload $w --steps 3 --no-cpu-baseline --save-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(python -c "import json;d=json.load(open('$O/bench_$w.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], 'frac', r.get('frac'), 'lanes', r.get('lane_utilisation'), 'hbm', r.get('hbm_measured_frac'), 'alg', r.get('algorithmic_GBps'))" 2>&1)"
done
timeout -k 10 600 python tools/partition_balance.py masterchief 5000 > $O/partition_balance.jsonl 2>&1; cat $O/partition_balance.jsonl | cut -c1-300
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --save-png $O/render_720p_5000spp.png > /dev/null 2>&1; ls -la $O/render_720p_5000spp.png
cp profiles/pmc_*.json $O/ 2>/dev/null
