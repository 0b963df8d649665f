#!/bin/bash
# round-2 batch 20: LDS-resident tree variant -- parity (both node paths), A/B per workload
set -o pipefail
O=gpurun_out/r2w; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "passed\|failed\|Error\|^E  " $O/pytest.txt | tail -8
for w in masterchief_720p_5000spp iron_720p_5000spp spheres_720p_1024spp sphere_field_720p_1024spp spheres_240p_64spp masterchief_1080p_8192spp; do
  for c in 1 0; do
    SRT_LDS_TREE=$c timeout -k 10 400 python bench.py --workload $w --steps 2 --no-cpu-baseline --no-pmc > $O/ab_${w}_$c.json 2>$O/ab_${w}_$c.err
    echo "$w lds_tree=$c: $(python -c "import json;d=json.load(open('$O/ab_${w}_$c.json'));print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel'], d['launch']['lds_bytes_per_workgroup'])" 2>&1)"
  done
done
