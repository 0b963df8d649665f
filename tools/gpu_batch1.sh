#!/bin/bash
# round-2 batch 1: parity suite, headline bench with in-run counters, step profile, soup points
set -o pipefail
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout -k 10 300 python tools/profile_steps.py masterchief 256 > $O/steps.txt 2>&1; tail -6 $O/steps.txt
timeout -k 10 600 python bench.py --steps 2 --save-pmc > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"; cut -c1-1500 $O/bench_headline.json; tail -3 $O/bench_headline.err
for w in soup_1m_720p_16spp soup_10m_720p_16spp soup_10m_ploc_closest_720p_16spp; do
  timeout -k 10 900 python bench.py --workload $w --steps 3 --no-cpu-baseline --save-pmc > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; cut -c1-1200 $O/bench_$w.json; tail -2 $O/bench_$w.err
done
cp profiles/pmc_*.json $O/ 2>/dev/null
