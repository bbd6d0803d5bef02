#!/bin/bash
# per-kernel PMC sums of any command: tools/pmc_any.sh "<COUNTER ...>" <kernel-name-filter> <cmd...>   (one --pmc pass; GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONPATH=$ROOT
CTRS=$1; FILT=$2; shift 2
OUT=/tmp/pmc_any_$$
cd $ROOT
timeout 600 rocprofv3 --output-format csv --pmc $CTRS -d $OUT -o p -- "$@" > /dev/null 2>&1
python - <<PY
import csv, glob
from collections import defaultdict
fs = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg, calls = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mpg::", "")
        if "$FILT" and "$FILT" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    n = len(calls[k])
    print("%-48s dispatches %d" % (k[:48], n), " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items())))
PY
rm -rf $OUT
