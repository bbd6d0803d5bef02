mkdir -p gpurun_out/r3r
timeout 1200 python -m pytest tests/test_gpu_gravity.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r3r/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3r/tests.log
tail -4 gpurun_out/r3r/tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3r/bench.json 2> gpurun_out/r3r/bench.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3r/bench.json") if x.startswith("{")][-1])
r=d["roofline"]; n=r["targets_per_launch"]
print("ms/step", d["ms_per_step"], "walk", r["avg_launch_ms"], "frac", r["frac"], "leaf entries/t", r["leaf_entries_per_launch"]/n, "nodes/t", r["node_entries_per_launch"]/n)
PY
