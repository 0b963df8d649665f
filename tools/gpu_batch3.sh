#!/bin/bash
# round-2 batch 3: new parity tests, chunk-count sweep, scheduler threshold sweep, other configs
set -o pipefail
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "published-image residuals\|passed\|failed\|Error\|assert" $O/pytest.txt | tail -15
SWEEP_CHUNKS=1,2,4,8,16,32,64,157,256 SWEEP_SHADE=16 SWEEP_PRIM=12 SWEEP_BURST=32 SWEEP_HIT=24 timeout -k 10 300 python tools/sweep.py masterchief 5000 > $O/sweep_chunks.txt 2>&1; cat $O/sweep_chunks.txt | tail -10
SWEEP_CHUNKS=63 SWEEP_SHADE=8,16,24,32 SWEEP_PRIM=8,12,16,24 SWEEP_BURST=32 SWEEP_HIT=16,24,32,40 timeout -k 10 600 python tools/sweep.py masterchief 2000 > $O/sweep_thresholds.txt 2>&1; sort -k12 -n -r $O/sweep_thresholds.txt | head -8; sort -k12 -n $O/sweep_thresholds.txt | head -3
for w in iron_720p_5000spp spheres_720p_1024spp spheres_240p_64spp; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$? $(python -c "import json;d=json.load(open('$O/bench_$w.json'));print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline'].get('frac'), d['roofline'].get('lane_utilisation'))" 2>&1)"
done
