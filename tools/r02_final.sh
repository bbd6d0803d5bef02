#!/bin/bash
# round 2, final GPU call: three full suites (no retries anywhere), the default bench line, the rocprofv3 profile of the same command
mkdir -p gpurun_out/final
export MASTER_ADDR=127.0.0.1
tools/flake_hunt.sh 3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"; tail -c 400 gpurun_out/final/bench_default.json
timeout 600 python bench.py --workload hydro --steps 5 --warmup 2 > gpurun_out/final/bench_hydro.json 2>/dev/null; echo "hydro rc=$?"
tools/prof.sh r02c_szel --no-extras > gpurun_out/final/prof.log 2>&1; tail -3 gpurun_out/final/prof.log
