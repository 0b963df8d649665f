#!/bin/bash
# round-2 batch 24: flattened shading records -- parity, headline + configs, old library side by side
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "passed\|failed\|Error\|^E  " $O/pytest.txt | tail -8
for w in masterchief_720p_5000spp spheres_720p_1024spp iron_720p_5000spp sphere_field_720p_1024spp spheres_240p_64spp soup_1m_720p_16spp; do
  for v in new old; do
    if [ $v = old ]; then export SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_before_shaderec.so; else unset SRT_HIP_LIB; fi
    timeout -k 10 400 python bench.py --workload $w --steps 2 --no-cpu-baseline --no-pmc > $O/ab_${w}_$v.json 2>$O/ab_${w}_$v.err
    echo "$w $v: $(python -c "import json;d=json.load(open('$O/ab_${w}_$v.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
  done
done
unset SRT_HIP_LIB
timeout -k 10 200 python tools/profile_steps.py masterchief 64 > $O/steps_masterchief.txt 2>&1; cat $O/steps_masterchief.txt
