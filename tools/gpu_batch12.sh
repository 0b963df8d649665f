#!/bin/bash
# round-2 batch 12: several primitive tests per scheduling trip (SRT_PRIM_ROUNDS), same-box A/B
set -o pipefail
O=gpurun_out/r2m; mkdir -p $O
for rep in 1 2; do
for v in "" prim1 prim2 prim3; do
  if [ -n "$v" ]; then export SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_$v.so; else unset SRT_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/${v}_$rep.json 2>/dev/null
  echo "'$v' rep $rep: $(python -c "import json;d=json.load(open('$O/${v}_$rep.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done
done
for v in prim2; do
  for am in 1 4 16 32; do
  SRT_PRIM_AGAIN_MIN=$am SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_$v.so timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/${v}_am$am.json 2>/dev/null
  echo "$v again_min $am: $(python -c "import json;d=json.load(open('$O/${v}_am$am.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
  done
done
SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_prim2.so timeout -k 10 300 python tools/profile_steps.py masterchief 256 2>&1 | tail -9
