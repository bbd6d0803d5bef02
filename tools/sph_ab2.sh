#!/bin/bash
# same-box A/B of library build variants on the hydro bench line (GPU box): tools/sph_ab2.sh <out> "<flags1>" "<flags2>" ...  ("" = default);
# flags apply to every source (the merge rule lives in ngb_walk.h, shared by sph.hip, fof.hip and grav_pair_walk.hip)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$1; shift
mkdir -p $(dirname $OUT); : > $OUT
for f in "$@"; do
    if [ -n "$f" ]; then export MPG_EXTRA_FLAGS="$f"; else unset MPG_EXTRA_FLAGS; fi
    python mp-gadget_amd/build.py > /dev/null 2>&1 || echo "build failed: $f" | tee -a $OUT
    for sph in de pe; do
    python bench.py --workload hydro --sph $sph --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json,re; j=json.loads(sys.stdin.read()); p=j['phases_ms']; r=j['roofline']; rh=j['roofline_hydro']
print('[%s %s] density %.3f ms hydro %.3f ms step %.2f ms | %s | %s' % ('$f', '$sph', p['density'], p['hydro'], j['ms_per_step'], re.search(r'\(\d+ neighbours[^)]*\)', r['note']).group(0), re.search(r'\(\d+ pairs[^)]*\)', rh['note']).group(0)))" | tee -a $OUT
    done
done
unset MPG_EXTRA_FLAGS
python mp-gadget_amd/build.py > /dev/null 2>&1
