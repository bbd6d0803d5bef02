mkdir -p gpurun_out/r3fin
bash tools/prof.sh r03d_final2 > /dev/null 2>&1
cp gpurun_out/prof_r03d_final2/walk_traffic.json profiles/walk_traffic.json
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r3fin/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3fin/tests.log
tail -4 gpurun_out/r3fin/tests.log
timeout 600 python bench.py > gpurun_out/r3fin/bench.json 2> gpurun_out/r3fin/bench.err; echo "bench rc=$?"
cp profiles/walk_traffic.json gpurun_out/r3fin/walk_traffic.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
