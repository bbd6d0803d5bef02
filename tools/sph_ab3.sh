#!/bin/bash
# same-box A/B of the SPH searches' leaf capacity (MPG_SPH_LEAF_CAP: 8 = the reference's leaves, 16 / 32 = search leaves of the round-5
# form, TreeBuilder::calc_search_links) on the hydro bench lines: tools/sph_ab3.sh <out> [caps...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$1; shift
CAPS=${*:-8 16}
mkdir -p $(dirname $OUT); : > $OUT
for cap in $CAPS; do
  for sph in de pe; do
    export MPG_SPH_LEAF_CAP=$cap
    python bench.py --workload hydro --sph $sph --steps 8 --warmup 2 --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys,json,re; j=json.loads(sys.stdin.read()); p=j['phases_ms']; r=j['roofline']; rh=j['roofline_hydro']
print('[leaf_cap=$cap %s] density %.3f ms hydro %.3f ms step %.2f ms | k_density %.3f ms | k_hydro %.3f ms | %s | %s' % ('$sph', p['density'], p['hydro'], j['ms_per_step'], r['avg_launch_ms'], rh['avg_launch_ms'], re.search(r'\(.*\)', r['note']).group(0)[:120], re.search(r'\(.*\)', rh['note']).group(0)[:80]))" | tee -a $OUT
  done
done
unset MPG_SPH_LEAF_CAP
