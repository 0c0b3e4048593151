# round 5, GPU call C: A/B of the diagonal-run loop variants (prefetch x in-place asm), pair_floor with the cost rule, the overlapped exchange
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
for v in "" _v01 _v10 _v00; do
  L=$R/rustqip_amd/lib/libqip_hip$v.so
  echo "{\"lib\": \"$v\"}" >> $O/diag_ab.jsonl
  QIP_HIP_LIB=$L timeout 300 python tools/bench_tile.py 30 5 qft,c4 1 >> $O/diag_ab.jsonl 2>> $O/err.txt
done
QIP_STATE_OPTS=pair_floor=1 timeout 900 python tools/bench_tile.py 30 3 c4,c2,qft 0 >> $O/gate_by_gate.jsonl 2>> $O/err.txt
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=8 -k "virtual_shards or pairs_a_line_floor or bench_multi_rank" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
