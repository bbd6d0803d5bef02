mkdir -p gpurun_out/r3t
cp mp-gadget_amd/libmpgadget_hip.so /tmp/lib_orig.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in NOARITH HALFLOAD; do
cp $R/tools/_bin/lib_$v.so $R/mp-gadget_amd/libmpgadget_hip.so
MPG_EXTRA_FLAGS="-DMPG_EXP_$v" rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/r3t/trace_$v -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3t/bench_$v.json 2> $R/gpurun_out/r3t/bench_$v.err
python - <<PY
import csv
print("$v")
for r in list(csv.DictReader(open("$R/gpurun_out/r3t/trace_$v/trace_kernel_stats.csv")))[:3]:
    print("  ", r["Name"][28:70], r["Calls"], "avg %.2f min %.2f max %.2f" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
PY
tail -2 $R/gpurun_out/r3t/bench_$v.err
done
cp /tmp/lib_orig.so $R/mp-gadget_amd/libmpgadget_hip.so
find $R/gpurun_out/r3t -name "*kernel_trace.csv" -delete
