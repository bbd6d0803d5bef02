#!/bin/bash
# per-dispatch durations of the fallback kernel (k_grav_walk_coop) in a bench run: tools/prof_coop_tail.sh [bench args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$ROOT
OUT=/tmp/prof_coop_$$
cd $ROOT
rocprofv3 --output-format csv --kernel-trace -d $OUT -o t -- python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | grep -o "ms_per_step[^,]*"
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_grav_walk_coop" in r["Kernel_Name"]]
print("fallback kernel dispatches (ms):", " ".join("%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows))
PY
rm -rf $OUT
