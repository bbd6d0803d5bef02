mkdir -p gpurun_out/r3x
timeout 1500 python -m pytest tests/test_gpu_sph.py tests/test_hydro_physics.py -x -q -m gpu > gpurun_out/r3x/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3x/tests.log
tail -5 gpurun_out/r3x/tests.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/r3x/trace -o trace -- python $R/bench.py --workload hydro --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3x/bench.json 2> $R/gpurun_out/r3x/bench.err
python - <<PY
import csv, json
for r in list(csv.DictReader(open("$R/gpurun_out/r3x/trace/trace_kernel_stats.csv")))[:8]:
    if "k_density" in r["Name"] or "k_hydro" in r["Name"]:
        print("  ", r["Name"][:40], r["Calls"], "avg %.2f min %.2f max %.2f" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
d=json.loads([x for x in open("$R/gpurun_out/r3x/bench.json") if x.startswith("{")][-1]); print(d["ms_per_step"], d.get("phases_ms"))
PY
find $R/gpurun_out/r3x -name "*kernel_trace.csv" -delete
