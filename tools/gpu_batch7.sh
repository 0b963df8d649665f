#!/bin/bash
# round-2 batch 7: same-box A/B (abtest/old = commit 4d01e91) of the two exact-chunk-sum paths, parity suite
set -o pipefail
O=gpurun_out/r2h; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "passed\|failed\|Error\|^E  " $O/pytest.txt | tail -12
for rep in 1 2; do
for w in masterchief_720p_5000spp spheres_240p_64spp spheres_720p_1024spp iron_720p_5000spp masterchief_1080p_8192spp; do
  (cd abtest/old && timeout -k 10 300 python bench.py --workload $w --steps 2 --no-cpu-baseline --no-pmc 2>/dev/null) > $O/old_${w}_$rep.json
  timeout -k 10 300 python bench.py --workload $w --steps 2 --no-cpu-baseline --no-pmc > $O/new_${w}_$rep.json 2>/dev/null
  echo "$w rep $rep: old $(python -c "import json;d=json.load(open('$O/old_${w}_$rep.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)   new $(python -c "import json;d=json.load(open('$O/new_${w}_$rep.json'));print(d['value'], d['roofline']['kernel_ms_avg'], d['config']['spp_chunks'])" 2>&1)"
done
done
