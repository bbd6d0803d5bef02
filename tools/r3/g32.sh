mkdir -p gpurun_out/r3ag
for cfg in "0 33554432" "1 8388608" "1 4194304" "1 2097152" "0 4194304"; do
set -- $cfg
MPG_SPLIT_OVERLAP=$1 MPG_SPLIT_SLICE=$2 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3ag/bench_$1_$2.json 2> gpurun_out/r3ag/bench_$1_$2.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3ag/bench_$1_$2.json") if x.startswith("{")][-1])
r=d["roofline"]
print("overlap $1 slice $2 ms/step", d["ms_per_step"], "walk", r["avg_launch_ms"], "frac", r["frac"])
PY
done
