#!/bin/bash
# round 6: the whole GPU suite on the current tree (output kept under gpurun_out/r06b)
out=gpurun_out/${1:-r06b}
mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -q --durations=25 ) > $out/gpu_tests.txt 2>&1
echo "rc=$?" >> $out/gpu_tests.txt
tail -n 45 $out/gpu_tests.txt
