# round 5, GPU call D: A/B of the diagonal-run loop variants (all four libraries built from the same sources), overlapped exchange v2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
for rep in 1 2; do
for v in "" _v01 _v10 _v00; do
  L=$R/rustqip_amd/lib/libqip_hip$v.so
  echo "{\"lib\": \"$v\"}" >> $O/diag_ab.jsonl
  QIP_HIP_LIB=$L timeout 300 python tools/bench_tile.py 30 5 qft,c4 1 >> $O/diag_ab.jsonl 2>> $O/err.txt
done
done
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q --durations=8 -k "virtual_shards" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -30 /tmp/dist_out.txt >> $O/tests.txt 2>/dev/null
