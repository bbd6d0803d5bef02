mkdir -p gpurun_out/r3q
for cfg in "0 1024" "1 1152" "2 1280" "4 1536"; do
set -- $cfg
MPG_LEAF_EXPAND=$1 MPG_LIST_CAP=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3q/bench_$1.json 2> gpurun_out/r3q/bench_$1.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3q/bench_$1.json") if x.startswith("{")][-1])
r=d["roofline"]; n=r["targets_per_launch"]
print("kx $1 walk", r["avg_launch_ms"], "pp/t", r["pp_interactions_per_launch"]/n, "used/t", r["nodes_used_per_launch"]/n, "leaf entries/t", r["leaf_entries_per_launch"]/n, "singles/t", r["single_source_entries_per_launch"]/n)
PY
done
