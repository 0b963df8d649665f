#!/bin/bash
# round-2 batch 8: bisect of the 1.2 % headline regression, FETCH_SIZE calibration for random 32-B gathers
set -o pipefail
O=gpurun_out/r2i; mkdir -p $O
for rep in 1 2; do
  (cd abtest/old && timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc 2>/dev/null) > $O/old_$rep.json
  echo "old rep $rep: $(python -c "import json;d=json.load(open('$O/old_$rep.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
  (cd abtest/old && timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc --spp-chunks 64 2>/dev/null) > $O/old64_$rep.json
  echo "old 64 chunks rep $rep: $(python -c "import json;d=json.load(open('$O/old64_$rep.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
  for v in "" var_SHORTCIRCUIT var_GENERICDIV var_BOTH; do
    if [ -n "$v" ]; then export SRT_HIP_LIB=sexy-raytracer_amd/csrc/exp/libsrt_$v.so; else unset SRT_HIP_LIB; fi
    timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/new_${v}_$rep.json 2>/dev/null
    echo "new '$v' rep $rep: $(python -c "import json;d=json.load(open('$O/new_${v}_$rep.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
  done
  unset SRT_HIP_LIB
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/ubench_gather tools/ubench_gather.hip && /tmp/ubench_gather > $O/gather.txt 2>&1; cat $O/gather.txt
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/gather_pmc -- /tmp/ubench_gather > $GRAFT_REPO_ROOT/$O/gather_pmc.txt 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r2i/gather_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gather" in r["Kernel_Name"]:
            kib = float(r["Counter_Value"]); n = 256 * 8 * 256 * 64
            print("FETCH_SIZE %.0f KiB for %d records of 32 B = %.1f B per record" % (kib, n, kib * 1024 / n))
PY
