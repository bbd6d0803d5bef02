mkdir -p gpurun_out/r3f
MPG_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29874 bench.py --gpus 4 --size 48 --ic s_clust --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r3f/p4.json 2> gpurun_out/r3f/p4.err
echo "rc=$?"; grep -o '"parity_check".*' gpurun_out/r3f/p4.json | cut -c1-1800
MPG_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29875 bench.py --gpus 2 --size 48 --ic s_zel --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r3f/p2.json 2> gpurun_out/r3f/p2.err
echo "rc=$?"; grep -o '"parity_check".*' gpurun_out/r3f/p2.json | cut -c1-1200
lscpu | head -25
for cfg in 1x1 1x8 1x16 8x16 16x8; do timeout 300 python tools/r3/cb_test.py 128 20 $cfg 2>&1 | tail -1; done
