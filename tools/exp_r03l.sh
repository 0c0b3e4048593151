cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "merged or parametrised or two_threads or compiled_at_run_time or jit_cache" --durations=5 > gpurun_out/r03l/pytest.txt 2>&1
tail -12 gpurun_out/r03l/pytest.txt
out=gpurun_out/r03l/bench_tile.jsonl
run() { env "$@" timeout 300 python tools/bench_tile.py 30 3 $CIRC $MODE >> $out 2>&1; }
CIRC=c2,c4,qft,grover MODE=1 run QIP_TILE_JIT=1
CIRC=c2,qft MODE=1 run QIP_TILE_JIT=3
CIRC=c2,c4 MODE=1 run QIP_TILE_JIT=1 QIP_TILE_RELABEL=1
CIRC=c2,c4,qft,grover MODE=2 run QIP_TILE_JIT=1 QIP_TILE_FMA=1
CIRC=c2,c4,qft,grover MODE=2 run QIP_TILE_JIT=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1
CIRC=c2,c4 MODE=2 run QIP_TILE_JIT=1 QIP_TILE_FMA=1 QIP_TILE_MERGE=1 QIP_TILE_RELABEL=1
python - <<'PY'
import json
for l in open("gpurun_out/r03l/bench_tile.jsonl"):
    try: d=json.loads(l)
    except Exception:
        print(l.strip()[:200]); continue
    print(d['circuit'],'tile',d['tile'],'jit',d['jit'],'relabel',d['relabel'],'fma',d['fma'],'merge',d.get('merge'),'sweeps',d['sweeps'],'ms',d['ms'],'norm-1',d['norm']-1)
PY
