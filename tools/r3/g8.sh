mkdir -p gpurun_out/r3h
MPG_FORCE_MGPU=1 MASTER_PORT=29871 timeout 600 python bench.py --gpus 1 --size 256 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3h/c4.json 2> gpurun_out/r3h/c4.err; echo rc=$?
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3h/c4.json") if x.startswith("{")][-1])
print("ms/step", d["ms_per_step"], "walk", d["roofline"]["avg_launch_ms"]); print(d["phases_ms"]); print(d.get("parity_check"))
PY
cd /tmp && export TMPDIR=/tmp
MPG_FORCE_MGPU=1 MASTER_PORT=29872 rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3h/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --size 256 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r3h/trace/trace_kernel_stats.csv")))
for r in rows[:30]: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e6,3), r["Percentage"])
PY
find gpurun_out/r3h -name '*.csv' -size +2M -delete
