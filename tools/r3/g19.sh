mkdir -p gpurun_out/r3s
tools/_bin/rsq_acc > gpurun_out/r3s/rsq_acc.txt 2>&1; cat gpurun_out/r3s/rsq_acc.txt
timeout 1200 python -m pytest tests/test_gpu_gravity.py -x -q -m gpu -k "walk_kernel_variants or walk_parity or probe or committed" > gpurun_out/r3s/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3s/tests.log
tail -3 gpurun_out/r3s/tests.log
for i in 1 2; do
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3s/bench.json 2> gpurun_out/r3s/bench.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3s/bench.json") if x.startswith("{")][-1])
r=d["roofline"]
print("ms/step", d["ms_per_step"], "walk", r["avg_launch_ms"], "frac", r["frac"])
PY
done
