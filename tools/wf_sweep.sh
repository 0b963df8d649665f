#!/bin/bash
# One-at-a-time sweep of the path-pool kernel's scheduling tunables on the headline frame (1000 spp), Msamples/s.
# usage (GPU box): bash tools/wf_sweep.sh [VAR=value ...  one run per argument] > gpurun_out/wf_sweep.txt
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --no-pmc --no-cpu-baseline --no-hbm-point --steps 3 --spp ${SPP:-1000} ${WORKLOAD:+--workload $WORKLOAD} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"; }
run X=default
for kv in "$@"; do run ${kv//,/ }; done
run X=default
