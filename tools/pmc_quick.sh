#!/bin/bash
# One PMC pass over the benchmark command; prints the trace kernel's counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --spp-per-step 32 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU -d $OUT/a -o pmc -- $BENCH > $OUT/a.log 2>&1 || { echo "pass a failed"; tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_IFETCH -d $OUT/b -o pmc -- $BENCH > $OUT/b.log 2>&1 || { echo "pass b failed"; tail -5 $OUT/b.log; }
python3 - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('gpurun_out/pmcq/*/*.db')):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'ptx_trace%' group by kernel_name, counter_name"): print(r)
PY
