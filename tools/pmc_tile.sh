# rocprofv3 PMC passes over the LDS-resident tile sweeps (tools/bench_tile.py); counters only, no API tracing.
# usage: bash tools/pmc_tile.sh [circuits] [modes] [n]     results under gpurun_out/pmc_tile{A,B}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-c2}; M=${2:-1,2}; N=${3:-28}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pmc_tileA -o a --output-format csv -- python $R/tools/bench_tile.py $N 1 $C $M > $R/gpurun_out/pmc_tileA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_tileB -o b --output-format csv -- python $R/tools/bench_tile.py $N 1 $C $M > $R/gpurun_out/pmc_tileB.log 2>&1
grep circuit $R/gpurun_out/pmc_tileB.log
