mkdir -p gpurun_out/r3g
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc
timeout 900 python -m pytest tests/test_gpu_gravity.py tests/test_gpu_bench.py -x -q -m gpu -k "walk_kernel_variants or list_kernels_agree or walk_parity or multi_gpu or probe" > gpurun_out/r3g/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3g/tests.log
tail -4 gpurun_out/r3g/tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --cpu-sample 2097152 > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r3g/bench.json") if x.startswith("{")][-1])
print("ms/step", d["ms_per_step"], "walk", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
c=d["cpu_baseline"]; print({k:c[k] for k in ("value","cores","processes","cgroup_cpu_limit","pairs_per_s_per_thread","walk_s_median_of_3","tree_build_own_share_s","tree_build_all_particles_s")})
PY
for thr in 65536 8192 1024; do
MPG_SPLIT_MIN_TARGETS=$thr timeout 300 python bench.py --workload substep --active-frac 0.001953125 --steps 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('thr $thr', {k:(v['all_particle_tree']['walk_ms'], v['all_particle_tree']['walk_kernel'], v['active_only_tree']['walk_ms']) for k,v in d['substeps'].items()})"
done
MPG_LISTS_MODE=2 bash tools/prof.sh r3g --no-extras > /dev/null 2>&1
grep -A9 "k_walk_lists8<false" gpurun_out/prof_r3g/summary.txt | grep -E "avg|VALU|SALU|BUSY" | head; grep "steady" gpurun_out/prof_r3g/summary.txt
