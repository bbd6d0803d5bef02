mkdir -p gpurun_out/r3m
timeout 1200 python -m pytest tests/test_gpu_gravity.py -x -q -m gpu > gpurun_out/r3m/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3m/tests.log
tail -4 gpurun_out/r3m/tests.log
for k in 0 1 2 4; do
MPG_LEAF_EXPAND=$k timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3m/bench_$k.json 2> gpurun_out/r3m/bench_$k.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3m/bench_$k.json") if x.startswith("{")][-1])
print("kx $k ms/step", d["ms_per_step"], "walk", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "cap", d["roofline"]["list_capacity"], "fallback", d["roofline"]["targets_to_fallback_kernel"])
PY
done
