mkdir -p gpurun_out/r3ah
for c in 2 1 3 4 8; do
MPG_SPLIT_CPW=$c python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3ah/bench_$c.json 2> gpurun_out/r3ah/bench_$c.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3ah/bench_$c.json") if x.startswith("{")][-1])
r=d["roofline"]
print("cpw $c ms/step", d["ms_per_step"], "walk", r["avg_launch_ms"], "frac", r["frac"])
PY
done
