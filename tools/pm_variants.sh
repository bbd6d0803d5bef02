#!/bin/bash
# same-box comparison of pm.hip build variants on the headline bench (GPU box): tools/pm_variants.sh "<flags1>" "<flags2>" ... ("" = default)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for f in "$@"; do
    if [ -n "$f" ]; then export MPG_EXTRA_FLAGS="pm.hip:$f"; else unset MPG_EXTRA_FLAGS; fi
    python mp-gadget_amd/build.py > /dev/null 2>&1 || echo "build failed: $f"
    for ic in ${ICS:-s_zel}; do
    python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-extras --ic $ic 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms']; print('[%s] %s step %.2f ms  pm %s  deposit %s  fft %s  transfer %s  readout %s' % ('$f', '$ic', j['ms_per_step'], p.get('pm_total'), p.get('pm_deposit'), p.get('pm_fft'), p.get('pm_transfer'), p.get('pm_readout')))"
    done
done
unset MPG_EXTRA_FLAGS
python mp-gadget_amd/build.py > /dev/null 2>&1
