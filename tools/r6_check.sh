#!/bin/bash
# (1) the data fixtures of tests/golden through the HIP path; (2) a -DQIP_HIP_TUNING build on the box: the GPU suite once more with the
# measured alternatives as options, which switches on the A/B identity asserts (tile form == direct form, row shapes, ...)
out=gpurun_out/${1:-r06x}
mkdir -p $out
timeout 600 python -m pytest tests/test_golden_fixtures.py -m gpu -q 2>&1 | tail -n 3
QIP_HIP_TUNING=1 python -m rustqip_amd.build > $out/tuning_build.log 2>&1; tail -n 1 $out/tuning_build.log | cut -c1-200
( time timeout 1700 python -m pytest tests -m gpu -q --durations=5 ) > $out/gpu_tests_tuning_build.txt 2>&1
echo "rc=$?" >> $out/gpu_tests_tuning_build.txt
tail -n 25 $out/gpu_tests_tuning_build.txt | cut -c1-300
