#!/bin/bash
# per-call durations (ms) of the kernels whose name contains $1, in launch order: tools/prof_calls.sh <substr> <cmd...>
PAT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$ROOT
OUT=/tmp/prof_calls_$$
cd $ROOT
rocprofv3 --output-format csv --kernel-trace -d $OUT -o t -- "$@" > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "$PAT" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("$PAT:", " ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows))
PY
rm -rf $OUT
