mkdir -p gpurun_out/r3aa
cp mp-gadget_amd/libmpgadget_hip.so /tmp/lib_orig.so
for v in base BAILOUT; do
[ "$v" != "base" ] && cp tools/_bin/lib_$v.so mp-gadget_amd/libmpgadget_hip.so
F=""; [ "$v" != "base" ] && F="grav_walk_split.hip:-DMPG_EXP_$v"
for ic in s_zel s_clust; do
MPG_EXTRA_FLAGS="$F" python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --ic $ic > gpurun_out/r3aa/bench_${v}_$ic.json 2> gpurun_out/r3aa/bench_${v}_$ic.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3aa/bench_${v}_$ic.json") if x.startswith("{")][-1])
r=d["roofline"]
print("$v $ic ms/step", d["ms_per_step"], "walk", r["avg_launch_ms"], "cap", r["list_capacity"], "fallback", r["targets_to_fallback_kernel"])
PY
done
done
cp /tmp/lib_orig.so mp-gadget_amd/libmpgadget_hip.so
