#!/bin/bash
# One documented GPU-box session script (replaces the per-batch transcripts of rounds 1-2).
# usage (through gpurun):  gpurun --timeout 900 -- 'bash tools/gpu_session.sh <outdir-name> <step> [<step> ...]'
# Steps run in order and the session stops at the first one that times out or is killed (never start a
# further GPU step after a hung one).  Output goes to gpurun_out/<outdir-name>/.
#   tests            python -m pytest tests -m gpu -x -q
#   bench            python bench.py (default headline line, counters in-run)
#   bench:<workload> python bench.py --workload <workload>
#   steps[:scene[:spp]]  tools/profile_steps.py (counting variant: share / fill / clocks per step kind)
#   stats            rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --no-pmc --no-cpu-baseline`
#   ab:<lib>:<workload>[:spp]  headline rate with SRT_HIP_LIB=<lib> (an experimental build under abtest/)
#   py:<script>[:args,...]     python <script> args...
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out/$1; shift
mkdir -p "$O"
export TMPDIR=/tmp
run() {  # run <seconds> <logfile> cmd...: stops the session when the command hits its time limit
  local lim=$1 log=$2; shift 2
  timeout -k 10 "$lim" "$@" > "$log" 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc $(basename "$log")"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out -- stopping the session"; exit 1; fi
  return $rc
}
for step in "$@"; do
  IFS=: read -r kind a1 a2 a3 <<< "$step"
  case $kind in
    tests) run 1100 "$O/pytest.txt" python -m pytest tests -m gpu -x -q; tail -3 "$O/pytest.txt";;
    bench) run 400 "$O/bench_${a1:-default}.json" python bench.py ${a1:+--workload $a1} ${a2:+--spp $a2}; cut -c1-600 "$O/bench_${a1:-default}.json";;
    steps) run 300 "$O/steps_${a1:-masterchief}.txt" python tools/profile_steps.py ${a1:-masterchief} ${a2:-64}; cat "$O/steps_${a1:-masterchief}.txt";;
    stats) (cd /tmp && run 400 "$O/stats.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-pmc --no-cpu-baseline)
           find "$O/stats" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \; ; head -5 "$O/kernel_stats.csv";;
    ab)    SRT_HIP_LIB=$ROOT/$a1 run 400 "$O/ab_$(basename $a1 .so)_$a2.json" python bench.py --workload $a2 ${a3:+--spp $a3} --no-pmc --no-cpu-baseline --steps 3
           cut -c1-200 "$O/ab_$(basename $a1 .so)_$a2.json";;
    py)    run 600 "$O/$(basename $a1 .py).txt" python $a1 ${a2//,/ }; tail -40 "$O/$(basename $a1 .py).txt";;
    *) echo "unknown step $step"; exit 2;;
  esac
done
