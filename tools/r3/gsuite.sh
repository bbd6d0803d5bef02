mkdir -p gpurun_out/r3suite
timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 > gpurun_out/r3suite/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3suite/tests.log
grep -E "passed|failed|FAILED|ERROR|tests rc" gpurun_out/r3suite/tests.log | tail -40
