cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "parametrised or jit_cache or two_threads or compiled_at_run_time or every_timed_leg" --durations=8 > gpurun_out/r03e/pytest.txt 2>&1
tail -30 gpurun_out/r03e/pytest.txt
for j in 1 2; do QIP_TILE_JIT=$j timeout 300 python tools/bench_tile.py 30 3 c2,qft 1 >> gpurun_out/r03e/bench_tile_param.jsonl 2>&1; done
cat gpurun_out/r03e/bench_tile_param.jsonl
