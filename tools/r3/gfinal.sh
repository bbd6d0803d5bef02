# full GPU suite + the driver's default bench line + the rocprofv3 passes the profiles/ summaries come from
mkdir -p gpurun_out/r3final
timeout 2700 python -m pytest tests/ -q -m gpu --durations=12 > gpurun_out/r3final/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3final/tests.log
grep -E "passed|failed|FAILED|ERROR|tests rc" gpurun_out/r3final/tests.log | tail -15
timeout 900 python bench.py > gpurun_out/r3final/bench_default.json 2> gpurun_out/r3final/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3final/bench_default.json") if x.startswith("{")][-1])
print("ms/step", d["ms_per_step"], "value", d["value"], "walk", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print({k:(v["ms_per_step"],v["walk_ms"]) for k,v in d["other_inputs"].items()}, "hydro", d["hydro"]["ms_per_step"], "host", d["host_path"]["ms_per_step"], "resident", d["resident_path"]["ms_per_step"])
c=d["cpu_baseline"]; print({k:c[k] for k in ("value","cores","processes","cgroup_cpu_limit","pairs_per_s_per_thread","value_if_all_physical_cores")})
PY
bash tools/prof.sh r03a_lists8 > /dev/null 2>&1
head -30 gpurun_out/prof_r03a_lists8/summary.txt
