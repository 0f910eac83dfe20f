#!/bin/bash
# PMC passes over the 10 M-triangle workload (GPU box): HBM bytes fetched / written and VALU counters of the
# scheduled kernel, one rocprofv3 run per counter group; summary into gpurun_out/prof_mesh/pmc_summary.txt.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_mesh
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --workload all_mesh --steps 2 --warmup 1 --no-cpu-baseline"
run_pmc() { local name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/$name.log; exit 1; }
  echo "pass $name done"
}
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run_pmc grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/prof_mesh/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('ptx_'):
            a = agg[(r['Kernel_Name'], r['Counter_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
lines = [f'{k[0]},{k[1]},sum={v[0]:.0f},dispatches={v[1]},per_dispatch={v[0]/max(v[1],1):.1f}' for k, v in sorted(agg.items())]
open('gpurun_out/prof_mesh/pmc_summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
