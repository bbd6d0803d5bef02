#!/bin/bash
# same-box comparison of sph.hip build variants (GPU box): tools/sph_variants.sh "<flags1>" "<flags2>" ...  ("" = the default build)
# prints the density / hydro phases of the hydro bench line per variant; the default build is restored at the end
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for f in "$@"; do
    if [ -n "$f" ]; then export MPG_EXTRA_FLAGS="sph.hip:$f"; else unset MPG_EXTRA_FLAGS; fi
    python mp-gadget_amd/build.py > /dev/null 2>&1 || echo "build failed: $f"
    python bench.py --workload hydro --steps 8 --warmup 2 --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms']; print('[%s] density %.3f hydro %.3f step %.2f' % ('$f', p['density'], p['hydro'], j['ms_per_step']))"
done
unset MPG_EXTRA_FLAGS
python mp-gadget_amd/build.py > /dev/null 2>&1
