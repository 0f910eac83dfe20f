#!/bin/bash
# One PMC pass over tools/perf_probe.py for one scene (GPU box): usage tools/pmc_probe.sh <scene> [PROBE_N]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export PROBE_N=${2:-2}
OUT=gpurun_out/pmcp
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/a -o pmc -- python3 tools/perf_probe.py "$1" > $OUT/a.log 2>&1 || { echo "pass failed"; tail -5 $OUT/a.log; exit 1; }
python3 - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('gpurun_out/pmcp/*/*.db')):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'ptx_%' group by kernel_name, counter_name"): print(r)
PY
