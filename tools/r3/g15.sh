mkdir -p gpurun_out/r3o
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 1024" "2 1280" "2 1536" "1 1152" "4 1536"; do
set -- $cfg
MPG_LEAF_EXPAND=$1 MPG_LIST_CAP=$2 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/r3o/trace_$1_$2 -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3o/bench_$1_$2.json 2> $R/gpurun_out/r3o/bench_$1_$2.err
python - <<PY
import csv, json
print("kx $1 cap $2")
for r in list(csv.DictReader(open("$R/gpurun_out/r3o/trace_$1_$2/trace_kernel_stats.csv")))[:4]:
    print("  ", r["Name"][28:70], r["Calls"], "avg %.2f min %.2f max %.2f" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
d=json.loads([x for x in open("$R/gpurun_out/r3o/bench_$1_$2.json") if x.startswith("{")][-1])
print("   ms/step", d["ms_per_step"], "walk", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "cap", d["roofline"]["list_capacity"], "fallback", d["roofline"]["targets_to_fallback_kernel"])
PY
done
find $R/gpurun_out/r3o -name "*kernel_trace.csv" -delete
