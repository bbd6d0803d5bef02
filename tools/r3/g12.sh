mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3l/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3l/tests.log
tail -4 gpurun_out/r3l/tests.log
timeout 600 python bench.py > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r3l/bench.json
