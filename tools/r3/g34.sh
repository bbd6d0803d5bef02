mkdir -p gpurun_out/r3w
cp mp-gadget_amd/libmpgadget_hip.so /tmp/lib_orig.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
rm -rf $R/gpurun_out/r3w/trace_$1
[ "$1" != "base" ] && cp $R/tools/_bin/lib_$1.so $R/mp-gadget_amd/libmpgadget_hip.so
MPG_EXTRA_FLAGS="$2" rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/r3w/trace_$1 -o trace -- python $R/bench.py --workload hydro --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3w/bench_$1.json 2> $R/gpurun_out/r3w/bench_$1.err
python - <<PY
import csv
print("$1")
for r in list(csv.DictReader(open("$R/gpurun_out/r3w/trace_$1/trace_kernel_stats.csv"))):
    if "k_density" in r["Name"] or "k_hydro" in r["Name"]:
        print("  ", r["Name"][:40], r["Calls"], "avg %.2f min %.2f max %.2f" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
PY
}
run base ""


cp /tmp/lib_orig.so $R/mp-gadget_amd/libmpgadget_hip.so
find $R/gpurun_out/r3w -name "*kernel_trace.csv" -delete
