mkdir -p gpurun_out/r3ab
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MPG_FORCE_MGPU=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/r3ab/trace -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check > $R/gpurun_out/r3ab/bench.json 2> $R/gpurun_out/r3ab/bench.err
tail -3 $R/gpurun_out/r3ab/bench.err
python - <<PY
import csv, glob, json
rows=[]
for f in glob.glob("$R/gpurun_out/r3ab/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob("$R/gpurun_out/r3ab/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY "+r.get("Direction","")))
rows.sort()
ev=[i for i,r in enumerate(rows) if "k_walk_eval" in r[2]]
i1=ev[-1]; i0=ev[-2]+1
t0=rows[i0][0]; busy_end=t0
for s,e,n in rows[i0:i1+1]:
    gap=(s-busy_end)/1e6
    if (e-s)/1e6 > 0.05 or gap > 0.05:
        print("%9.3f %9.3f dur %8.3f gap %7.3f %s" % ((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,gap,n))
    busy_end=max(busy_end,e)
d=json.loads([x for x in open("$R/gpurun_out/r3ab/bench.json") if x.startswith("{")][-1]); print(d["ms_per_step"], d["phases_ms"])
PY
find $R/gpurun_out/r3ab -name "*trace.csv" -size +4M -delete
