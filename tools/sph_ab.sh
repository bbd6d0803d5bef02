#!/bin/bash
# same-box A/B of the SPH searches: cubes around the nodes' particles (default) against the reference's cell test (MPG_SPH_CELL_CULL=1)
# on the hydro bench lines (2 x 128^3 density-entropy, and with PE=1 the pressure-entropy form): tools/sph_ab.sh <out>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$1
mkdir -p $(dirname $OUT); : > $OUT
for cull in 0 1; do
  for sph in de pe; do
    if [ $cull = 1 ]; then export MPG_SPH_CELL_CULL=1; else unset MPG_SPH_CELL_CULL; fi
    python bench.py --workload hydro --sph $sph --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json,re; j=json.loads(sys.stdin.read()); p=j['phases_ms']; r=j['roofline']; rh=j['roofline_hydro']
print('[cell_cull=$cull %s] density %.3f ms hydro %.3f ms step %.2f ms | k_density %.3f ms frac %.4f | k_hydro %.3f ms frac %.4f | %s | %s' % ('$sph', p['density'], p['hydro'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], rh['avg_launch_ms'], rh['frac'], re.search(r'\(.*\)', r['note']).group(0)[:120], re.search(r'\(.*\)', rh['note']).group(0)[:80]))" | tee -a $OUT
  done
done
unset MPG_SPH_CELL_CULL
