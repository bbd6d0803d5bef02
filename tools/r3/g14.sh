mkdir -p gpurun_out/r3n
timeout 1200 python -m pytest tests/test_gpu_gravity.py -x -q -m gpu > gpurun_out/r3n/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3n/tests.log
tail -4 gpurun_out/r3n/tests.log
cd /tmp && export TMPDIR=/tmp
for k in 0 2; do
MPG_LEAF_EXPAND=$k rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3n/trace_$k -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r3n/bench_$k.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3n/bench_$k.err
echo "kx $k"; find $GRAFT_REPO_ROOT/gpurun_out/r3n/trace_$k -name "*kernel_stats.csv" | xargs head -6 | cut -c1-160
done
find $GRAFT_REPO_ROOT/gpurun_out/r3n -name "*kernel_trace.csv" -delete
