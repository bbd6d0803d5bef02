#!/bin/bash
# calibration of FETCH_SIZE / WRITE_SIZE on the walk kernels' access patterns (GPU box): tools/pmc_calib.sh <outdir>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/pmc_calib}
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/pmc_calib.hip -o $OUT/pmc_calib || exit 1
cd /tmp && export TMPDIR=/tmp
$OUT/pmc_calib 4 > $OUT/bytes.txt
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- $OUT/pmc_calib 4 > /dev/null 2> $OUT/fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o pmc -- $OUT/pmc_calib 4 > /dev/null 2> $OUT/write.err
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
B = 4 << 30
known = {"k_read16": B, "k_read4": B, "k_read4g": B, "k_gather32": B, "k_write4": B, "k_write4run": B}
lines = [open(os.path.join(out, "bytes.txt")).read().strip()]
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if r["Counter_Name"] == name and k in known:
                v = float(r["Counter_Value"]) * 1024
                lines.append("%-10s %-12s reported %.3f GB for %.3f GB moved: factor %.3f" % (name, k, v / 1e9, known[k] / 1e9, v / known[k]))
open(os.path.join(out, "calibration.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -f $OUT/pmc_calib
find $OUT -name '*.csv' -size +1M -delete
