# round 5, GPU call A: what changed since r4 on the GPU side — diagonal runs in the interpreter, helper-process compilation + disk
# cache, tile_auto programs (wide tiles inside hipGraphs), ADVICE fixes on the sharded path, dense3_inline — tests, then timings.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
nproc > $O/env.txt; cat /sys/fs/cgroup/cpu.max >> $O/env.txt 2>&1
K="lds_tile_multi_gate or compiled_at_run_time or complex64_and_programs or hipgraph_program or programs_compile or fuzz_every_gate or parametrised or jit_cache_is_bounded or two_threads or relabelled_layout_persists or wide_tiles or virtual_shards or oracle_windows_complex64 or every_timed_leg or oracle_windows and 30 or builder_run_loop or tile_relabel_is_bit or config_circuits or qft_matches or tile2_merged"
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=30 -k "$K" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
python - > $O/jit_counters.txt 2>&1 <<'PY'
from rustqip_amd import _ffi
print(_ffi.jit_counters(), _ffi.lib.qip_hip_jit_cache_dir())
PY
# interpreter sweeps with / without the diagonal runs (n = 30, median-ish: best of 3)
for dr in 1 0; do QIP_TILE_DIAG_RUNS=$dr timeout 600 python tools/bench_tile.py 30 3 qft,c4,c2,grover 1 >> $O/tile_interp.jsonl 2>> $O/err.txt; done
QIP_TILE_DIAG_RUNS=1 timeout 300 python tools/bench_tile.py 30 3 qft,c4 2 >> $O/tile_interp.jsonl 2>> $O/err.txt
# dense-k3 Grover on wide tiles: pass_dense3w vs the groups written out
for inl in 0 1; do QIP_TILE_JIT=1 QIP_TILE_WIDE=1 QIP_TILE_WIDE_DENSE3_INLINE=$inl timeout 600 python tools/bench_tile.py 30 3 groverk3 1 >> $O/tile_k3.jsonl 2>> $O/err.txt; done
QIP_TILE_JIT=1 timeout 600 python tools/bench_tile.py 30 3 groverk3 1 >> $O/tile_k3.jsonl 2>> $O/err.txt
# cold compile time of the wide relabelled plan with helpers (fresh box: nothing cached) and a second process (disk cache)
( time QIP_TILE_JIT=1 QIP_TILE_WIDE=1 QIP_TILE_RELABEL=1 python tools/bench_tile.py 30 3 c2 1 ) >> $O/tile_wide.jsonl 2>> $O/time_wide.txt
( time QIP_TILE_JIT=1 QIP_TILE_WIDE=1 QIP_TILE_RELABEL=1 python tools/bench_tile.py 30 3 c2 1 ) >> $O/tile_wide.jsonl 2>> $O/time_wide.txt
ls ~/.cache/qip_hip | wc -l >> $O/env.txt
