mkdir -p gpurun_out/r3ad
cp mp-gadget_amd/libmpgadget_hip.so /tmp/lib_orig.so
for v in "$@"; do
n=${v%%:*}; f=${v#*:}
cp tools/_bin/lib_$n.so mp-gadget_amd/libmpgadget_hip.so
MPG_EXTRA_FLAGS="grav_walk_split.hip:$f" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3ad/bench_$n.json 2> gpurun_out/r3ad/bench_$n.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3ad/bench_$n.json") if x.startswith("{")][-1])
r=d["roofline"]
print("$n ms/step", d["ms_per_step"], "walk", r["avg_launch_ms"], "frac", r["frac"])
PY
done
cp /tmp/lib_orig.so mp-gadget_amd/libmpgadget_hip.so
