#!/bin/bash
# two processes in a row on a fresh cache directory: the second one must take the compiled sweeps
export QIP_HIP_CACHE_DIR=/tmp/qip_dbg_cache_$$
n=${1:-26}
python tools/builder_one_shot.py $n
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 1; echo "t=$i: $(ls $QIP_HIP_CACHE_DIR | grep -c '\.co$') .co, $(ls $QIP_HIP_CACHE_DIR | grep -c '^seg') seg, $(ls $QIP_HIP_CACHE_DIR | grep -vc '\.co$') other"; done
ls -la $QIP_HIP_CACHE_DIR | head -30
python tools/builder_one_shot.py $n
sleep 6
ls $QIP_HIP_CACHE_DIR | wc -l
python tools/builder_one_shot.py $n
