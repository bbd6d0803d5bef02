#!/bin/bash
# kernel timeline of the LAST step of a bench run (start offset and duration of every kernel between the last two walk evaluations):
# tools/timeline.sh [bench args]   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$ROOT
OUT=/tmp/timeline_$$
cd $ROOT
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $OUT -o t -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-parity-check "$@" > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [i for i, r in enumerate(rows) if "k_walk_eval<" in r["Kernel_Name"]]
# the timed steps come before the instrumented passes: take the step between the 3rd and 4th evaluation kernels
a, b = ev[2], ev[3]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mpg::", "").split("(")[0][:60]
    print("%9.3f ms  +%8.3f ms  stream %s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Stream_Id", r.get("Queue_Id", "?")), n))
PY
rm -rf $OUT
