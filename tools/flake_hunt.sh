#!/bin/bash
# Repeat the whole -m gpu suite (no retries) and keep every log: used to chase the rare multi-rank failure.
# usage: tools/flake_hunt.sh <repeats> [extra pytest args]
n=${1:-6}; shift
mkdir -p gpurun_out/flake
for i in $(seq 1 $n); do
  python -m pytest tests -m gpu -q --tb=short --durations=8 -p no:cacheprovider "$@" > gpurun_out/flake/run_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/flake/run_$i.log)"
done
grep -l "FAILED\|failed" gpurun_out/flake/run_*.log
