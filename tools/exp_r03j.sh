cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
timeout 300 python tools/exp_r03i.py > gpurun_out/r03j/h_sweep_modes.txt 2>&1
cat gpurun_out/r03j/h_sweep_modes.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "persists" 2>&1 | tail -3
