#!/bin/bash
# after `tools/final_bench.sh` + `tools/prof.sh r06_final` on the GPU box (their output merged into gpurun_out/): copy what is judged into
# profiles/ and print the numbers the documents quote.  Run in the build container, from the repository root.
set -e
cp gpurun_out/r06f/bench_r06_*.json profiles/
cp gpurun_out/prof_r06_final/summary.txt gpurun_out/prof_r06_final/walk_traffic.json gpurun_out/prof_r06_final/sph_traffic.json \
   gpurun_out/prof_r06_final/bench_trace.json gpurun_out/prof_r06_final/bench_hydro.json gpurun_out/r06f/gpu_suite.log profiles/r06_final/
cp "$(find gpurun_out/prof_r06_final/trace -name '*kernel_stats.csv' | head -1)" profiles/r06_final/kernel_stats.csv
cp "$(find gpurun_out/prof_r06_final/trace_hydro -name '*kernel_stats.csv' | head -1)" profiles/r06_final/kernel_stats_hydro.csv
cp gpurun_out/prof_r06_final/walk_traffic.json gpurun_out/prof_r06_final/sph_traffic.json profiles/
python3 - <<'PY'
import json
def line(f): return json.loads(open('profiles/bench_r06_%s.json' % f).read().strip().splitlines()[-1])
print('stamps:', json.load(open('profiles/walk_traffic.json'))['build_stamp'], open('mp-gadget_amd/libmpgadget_hip.so.stamp').read())
for f in ['256_szel', '256_szel_1rank_rccl', '512_szel_1gpu', 'hydro_2x128', 'hydro_2x256_pe_1gpu']:
    d = line(f); r = d['roofline']
    print(f, 'ms/step %.2f' % d['ms_per_step'], 'value %.4g' % d['value'], 'frac %.4f' % r['frac'], r.get('kernels_ms'), 'launch ms %.2f' % r['avg_launch_ms'], d.get('phases_ms'))
d = line('256_szel'); hp = d['host_path']
print('host_path', hp['ms_per_step'], 'prefetched', hp['prefetched']['ms_per_step'], 'synchronous', hp['synchronous']['ms_per_step'])
print('other inputs', {k: (v['ms_per_step'], v['walk_ms']) for k, v in d['other_inputs'].items()})
print('cpu_baseline %.4g' % d['cpu_baseline']['value'], 'ratio %.0f' % (d['value'] / d['cpu_baseline']['value']), 'traffic %.4g' % d['roofline']['traffic'])
for w in ['integrate', 'fof', 'domain', 'substep']:
    d = line(w); print(w, 'ms %.3f' % d['ms_per_step'], 'value %.4g' % d['value'])
for k, v in line('substep')['substeps'].items():
    print(' substep', k, v['all_particle_tree']['ms_per_substep'], v['active_only_tree']['ms_per_substep'])
PY
grep -n "headline walks" profiles/r06_final/summary.txt; tail -1 profiles/r06_final/gpu_suite.log
