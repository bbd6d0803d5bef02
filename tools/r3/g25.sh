mkdir -p gpurun_out/r3y
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $R/gpurun_out/r3y/trace -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r3y/bench.json 2> $R/gpurun_out/r3y/bench.err
python - <<PY
import csv, glob
rows=[]
for f in glob.glob("$R/gpurun_out/r3y/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob("$R/gpurun_out/r3y/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY "+r.get("Direction","")))
rows.sort()
# last step: find last k_walk_eval, go back to previous k_walk_eval end
ev=[i for i,r in enumerate(rows) if "k_walk_eval" in r[2]]
i1=ev[-1]; i0=ev[-2]+1
t0=rows[i0][0]
prev_end=rows[i0-1][1]
print("gap before step start %.3f ms" % ((rows[i0][0]-prev_end)/1e6))
busy_end=rows[i0][0]
for s,e,n in rows[i0:i1+1]:
    gap=(s-busy_end)/1e6
    print("%9.3f %9.3f dur %8.3f gap %7.3f %s" % ((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,gap,n))
    busy_end=max(busy_end,e)
PY
find $R/gpurun_out/r3y -name "*trace.csv" -size +4M -delete
