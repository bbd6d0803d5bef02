mkdir -p gpurun_out/r3z
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r3z/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3z/tests.log
tail -4 gpurun_out/r3z/tests.log
timeout 600 python bench.py > gpurun_out/r3z/bench.json 2> gpurun_out/r3z/bench.err; echo "bench rc=$?"
