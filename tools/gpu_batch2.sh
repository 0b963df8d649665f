#!/bin/bash
# round-2 batch 2: K2 parity, VALU issue-rate microbenchmark, A/B of kernel variants and timing probes
set -o pipefail
O=gpurun_out/r2c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -3 $O/pytest.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/ubench_valu tools/ubench_valu.hip > $O/ubench_build.txt 2>&1 && timeout -k 10 120 /tmp/ubench_valu > $O/ubench.txt 2>&1; cat $O/ubench.txt
timeout -k 10 300 python tools/profile_steps.py masterchief 256 > $O/steps.txt 2>&1; tail -9 $O/steps.txt
for v in k1 k2 probe_LOAD128 probe_LOAD32 probe_VALU; do
  SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_$v.so timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/bench_$v.json 2> $O/bench_$v.err
  echo "$v rc=$? $(python -c "import json;d=json.load(open('$O/bench_$v.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done
timeout -k 10 600 python bench.py --steps 2 --no-cpu-baseline --save-pmc > $O/bench_k2_pmc.json 2> $O/bench_k2_pmc.err; echo "pmc rc=$?"; python -c "import json;d=json.load(open('$O/bench_k2_pmc.json'));print(d['value'], json.dumps(d['roofline'])[:900])"
