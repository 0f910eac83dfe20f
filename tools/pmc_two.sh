#!/bin/bash
# One PMC pass over tools/march_tasks_probe.py (both kernels run in it): instruction counts and wait cycles per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -d $OUT/a -o pmc -- python3 tools/march_tasks_probe.py > $OUT/a.log 2>&1 || { echo "pass failed"; tail -5 $OUT/a.log; exit 1; }
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
for db in sorted(glob.glob(sys.argv[1] + '/*/*.db')):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'ptx_sched%' group by kernel_name, counter_name"): print(r)
PY
