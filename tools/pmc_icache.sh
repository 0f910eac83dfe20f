#!/bin/bash
# Instruction-fetch counters of the trace kernel for one scene (GPU box): usage tools/pmc_icache.sh <scene> [samples per launch]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export PROBE_N=${2:-16} PROBE_REPEAT=1
OUT=gpurun_out/pmci_$1
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $OUT/a -o pmc -- python3 tools/perf_probe.py "$1" > $OUT/a.log 2>&1 || { echo "pass failed"; tail -5 $OUT/a.log; exit 1; }
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
for db in sorted(glob.glob(sys.argv[1] + '/*/*.db')):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'ptx_%' group by kernel_name, counter_name"): print(r)
PY
