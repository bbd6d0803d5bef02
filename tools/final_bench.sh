#!/bin/bash
# the bench lines committed under profiles/ at the end of a round (run on the GPU box through gpurun)
mkdir -p gpurun_out/r06f
python bench.py > gpurun_out/r06f/bench_r06_256_szel.json 2> gpurun_out/r06f/e1.txt
python bench.py --size 512 --no-cpu-baseline --no-extras > gpurun_out/r06f/bench_r06_512_szel_1gpu.json 2> gpurun_out/r06f/e2.txt
python bench.py --workload hydro > gpurun_out/r06f/bench_r06_hydro_2x128.json 2> gpurun_out/r06f/e3.txt
python bench.py --workload hydro --size 256 --sph pe > gpurun_out/r06f/bench_r06_hydro_2x256_pe_1gpu.json 2> gpurun_out/r06f/e4.txt
MPG_FORCE_MGPU=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --gpus 1 --no-cpu-baseline > gpurun_out/r06f/bench_r06_256_szel_1rank_rccl.json 2> gpurun_out/r06f/e5.txt
tail -c 600 gpurun_out/r06f/e*.txt; wc -c gpurun_out/r06f/*.json
