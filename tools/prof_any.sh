#!/bin/bash
# kernel-time summary (rocprofv3 --kernel-trace --stats) of any command run from the repo root: tools/prof_any.sh <cmd...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$ROOT
OUT=/tmp/prof_any_$$
cd $ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- "$@" 2>/dev/null | grep -v "^$" | tail -5
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-72s %5s calls %9.2f ms total %8.3f ms avg" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
rm -rf $OUT
