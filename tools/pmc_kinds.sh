# PMC: vector/scalar instructions per tile sweep, one gate kind per sweep (tools/probe_tile.py N kinds)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-28}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_kinds -o k --output-format csv -- python $R/tools/probe_tile.py $N kinds > $R/gpurun_out/pmc_kinds.log 2>&1
grep case $R/gpurun_out/pmc_kinds.log
