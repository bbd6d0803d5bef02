cp mp-gadget_amd/libmpgadget_hip.so /tmp/lib_orig.so
cp tools/_bin/lib_SPHCYC.so mp-gadget_amd/libmpgadget_hip.so
MPG_EXTRA_FLAGS="sph.hip:-DMPG_EXP_SPH_CYCLES" python bench.py --workload hydro --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "k_hydro per wave" | tail -2
cp /tmp/lib_orig.so mp-gadget_amd/libmpgadget_hip.so
