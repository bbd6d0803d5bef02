mkdir -p gpurun_out/r3v
cp mp-gadget_amd/libmpgadget_hip.so /tmp/lib_orig.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name flags
cp $R/tools/_bin/lib_$1.so $R/mp-gadget_amd/libmpgadget_hip.so
MPG_EXTRA_FLAGS="$2" rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/r3v/trace_$1 -o trace -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3v/bench_$1.json 2> $R/gpurun_out/r3v/bench_$1.err
python - <<PY
import csv, json
print("$1")
for r in list(csv.DictReader(open("$R/gpurun_out/r3v/trace_$1/trace_kernel_stats.csv")))[:2]:
    print("  ", r["Name"][28:70], r["Calls"], "avg %.2f min %.2f max %.2f" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
try:
    d=json.loads([x for x in open("$R/gpurun_out/r3v/bench_$1.json") if x.startswith("{")][-1]); print("   walk", d["roofline"]["avg_launch_ms"], "accel", d["resident_path"]["mean_abs_accel"] if "resident_path" in d else "")
except Exception as e: print("   bench failed", e)
PY
}
for v in XVALU XLDS XLOAD; do run $v "grav_walk_split.hip:-DMPG_EXP_$v"; done
cp /tmp/lib_orig.so $R/mp-gadget_amd/libmpgadget_hip.so
find $R/gpurun_out/r3v -name "*kernel_trace.csv" -delete
