#!/bin/bash
# Collects the round's rocprofv3 evidence for the benchmark command on a GPU box.  Each pass is its own
# run (kernel trace + stats; then one PMC pass per counter group), written under gpurun_out/prof/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --spp-per-step 32 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU -d $OUT/pmc_sq1 -o pmc -- $BENCH > $OUT/pmc_sq1.log 2>&1 || { echo "pmc sq1 failed"; tail -5 $OUT/pmc_sq1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1 || { echo "pmc sq2 failed"; tail -5 $OUT/pmc_sq2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_grbm -o pmc -- $BENCH > $OUT/pmc_grbm.log 2>&1 || { echo "pmc grbm failed"; tail -5 $OUT/pmc_grbm.log; exit 1; }
find $OUT -name "*.csv" | head -40
