#!/bin/bash
# same-box comparison of grav_walk_split.hip build variants (GPU box): tools/lists_variants.sh <out> "<flags1>" "<flags2>" ...
# ("" = the default build; a leading "F32=0 " in a variant keeps the fp64 node tests for its run, otherwise MPG_LISTS_F32=1); prints step, walk and the two kernels' times.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for v in "$@"; do
    f=$v; envf=1
    if [[ "$v" == F32=0* ]]; then envf=0; f=${v#F32=0}; f=${f# }; fi
    if [ -n "$f" ]; then export MPG_EXTRA_FLAGS="grav_walk_split.hip:$f"; else unset MPG_EXTRA_FLAGS; fi
    python mp-gadget_amd/build.py > /dev/null 2>&1 || echo "build failed: $f" | tee -a $OUT
    for ic in ${ICS:-s_zel}; do
        MPG_LISTS_F32=$envf python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --ic $ic 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; k=r.get('kernels_ms',{}); print('[%s] %s step %.2f ms  walk %.2f ms  lists %s eval %s  fp64_passes %s' % ('$v', '$ic', j['ms_per_step'], r['avg_launch_ms'], k.get('k_walk_lists8'), k.get('k_walk_eval'), r.get('fp32_fallback_passes_per_launch')))" | tee -a $OUT
    done
done
unset MPG_EXTRA_FLAGS
python mp-gadget_amd/build.py > /dev/null 2>&1
