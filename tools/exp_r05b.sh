# round 5, GPU call B: diagonal-run loop v2 (64-byte steps, element mask, in-place products), pair_floor, dense3_inline default on
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
K="pairs_a_line_floor or lds_tile_multi_gate or complex64_and_programs or hipgraph_program or programs_compile or fuzz_every_gate or wide_tiles or config_circuits or qft_matches or builder_run_loop or oracle_windows and 28 or selectors_inside or controlled_single or diagonal_gates or relabelled_layout"
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=12 -k "$K" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
QIP_TILE_DIAG_RUNS=1 timeout 600 python tools/bench_tile.py 30 5 qft,c4,c2 1 >> $O/tile_interp.jsonl 2>> $O/err.txt
for pf in 1 0; do QIP_STATE_OPTS=pair_floor=$pf timeout 900 python tools/bench_tile.py 30 3 c4,c2,qft 0 >> $O/gate_by_gate.jsonl 2>> $O/err.txt; done
QIP_TILE_JIT=1 QIP_TILE_WIDE=1 timeout 600 python tools/bench_tile.py 30 5 groverk3,grover 1 >> $O/tile_k3.jsonl 2>> $O/err.txt
