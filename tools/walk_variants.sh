#!/bin/bash
# same-box comparison of grav_walk_split.hip build variants on the headline bench (GPU box): tools/walk_variants.sh "<flags1>" "<flags2>" ...
# ("" = the default build); prints ms per step and the walk's phases; the default build is restored at the end
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for f in "$@"; do
    if [ -n "$f" ]; then export MPG_EXTRA_FLAGS="grav_walk_split.hip:$f"; else unset MPG_EXTRA_FLAGS; fi
    python mp-gadget_amd/build.py > /dev/null 2>&1 || echo "build failed: $f"
    python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline ${BENCH_EXTRA:-} 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('[%s] step %.2f ms  frac %.4f  walk %s  leaf entries %s' % ('$f', j['ms_per_step'], r['frac'], j.get('phases_ms',{}).get('walk'), r.get('leaf_entries_per_launch')))"
done
unset MPG_EXTRA_FLAGS
python mp-gadget_amd/build.py > /dev/null 2>&1
