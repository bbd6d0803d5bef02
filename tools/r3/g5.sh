set -x
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests/test_gpu_bench.py tests/test_gpu_cabi.py tests/test_hydro_physics.py -x -q -m gpu --durations=8 > gpurun_out/r3e/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3e/tests.log
tail -30 gpurun_out/r3e/tests.log
timeout 900 python bench.py > gpurun_out/r3e/bench_default.json 2> gpurun_out/r3e/bench_default.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/r3e/bench_default.json
tail -5 gpurun_out/r3e/bench_default.err
