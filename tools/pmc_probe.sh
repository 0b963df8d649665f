#!/bin/bash
# Extra hardware-counter passes over one render-kernel launch (bench.py --pmc-child), one rocprofv3 run per group.
# usage (on the GPU box): bash tools/pmc_probe.sh <outdir> <workload> <spp> "<counters of group 1>" ["<group 2>" ...]
# Prints, per group, the counters of the LAST render-kernel dispatch summed over its rows.  A group with a counter the
# device does not know fails by itself; the others still run.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$1; W=$2; SPP=$3; shift 3
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp || exit 1
[ -f "$O/counters_available.txt" ] || timeout -k 10 120 rocprofv3 -L > "$O/counters_available.txt" 2>&1
g=0
for group in "$@"; do
  g=$((g + 1))
  d=$O/pmc_$g
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$d" -- python3 "$ROOT/bench.py" --pmc-child --workload "$W" --spp "$SPP" > "$O/pmc_$g.log" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "group $g timed out -- stopping"; exit 1; fi
  if [ $rc -ne 0 ]; then echo "group $g ($group) failed rc=$rc: $(tail -2 "$O/pmc_$g.log")"; continue; fi
  python3 - "$d" <<'EOF'
import csv, glob, sys
per = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "srt_render_" not in row["Kernel_Name"]:
            continue
        v = per.setdefault(int(row.get("Dispatch_Id", 0) or 0), {})
        v[row["Counter_Name"]] = v.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
if per:
    for k, v in sorted(per[max(per)].items()):
        print("%-36s %.6g" % (k, v))
EOF
done
