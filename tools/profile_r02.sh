#!/bin/bash
# rocprofv3 evidence for one bench workload on a GPU box: kernel trace + stats, then one PMC pass per counter group
# (each its own run).  usage: tools/profile_r02.sh <workload> [extra bench args]; summaries land in gpurun_out/prof_<workload>/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
W=${1:-cornell}; shift || true
OUT=gpurun_out/prof_$W
rm -rf $OUT; mkdir -p $OUT
STEPS=3; [ "$W" = fractals ] && STEPS=2
BENCH="python3 bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
run_pmc() { # name counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
run_pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU
run_pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - "$W" "$OUT" <<'PY'
import csv, glob, json, collections, sys
W, OUT = sys.argv[1], sys.argv[2]
rows = []
for f in sorted(glob.glob(f'{OUT}/trace/**/*kernel_stats.csv', recursive=True)):
    rows += [r for r in csv.DictReader(open(f)) if r['Name'].startswith('ptx_')]
open(f'{OUT}/kernel_stats_summary.txt', 'w').write('\n'.join(','.join(f'{k}={v}' for k, v in r.items()) for r in rows) + '\n')
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f'{OUT}/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith(('ptx_trace', 'ptx_sched')):
            a = agg[(r['Kernel_Name'], r['Counter_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
lines = [f'{k[0]},{k[1]},sum={v[0]:.0f},dispatches={v[1]},per_dispatch={v[0]/max(v[1],1):.1f}' for k, v in sorted(agg.items())]
g = {k[1]: v[0] / max(v[1], 1) for k, v in agg.items()}
derived = {}
if 'GRBM_GUI_ACTIVE' in g and 'SQ_ACTIVE_INST_VALU' in g:
    simd_cycles = g['GRBM_GUI_ACTIVE'] / 8 * 1024  # the counter sums 8 XCDs; 256 CUs x 4 SIMDs
    derived = {"valu_busy_frac": round(g['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles, 4),
               "lane_utilisation": round(g['SQ_THREAD_CYCLES_VALU'] / (g['SQ_ACTIVE_INST_VALU'] * 64), 4),
               "salu_per_valu": round(g.get('SQ_INSTS_SALU', 0) / max(g['SQ_INSTS_VALU'], 1), 4),
               # instructions issued per SIMD cycle (a SIMD issues a wave64 v_fma_f32 every 2.5 cycles at best, a scalar instruction every 4.3: tools/probes/issue_probe.hip)
               "valu_per_simd_cycle": round(g['SQ_INSTS_VALU'] / simd_cycles, 4), "salu_per_simd_cycle": round(g.get('SQ_INSTS_SALU', 0) / simd_cycles, 4),
               "lds_bank_conflict_frac": round(g.get('SQ_LDS_BANK_CONFLICT', 0) / max(g.get('SQ_ACTIVE_INST_LDS', 1), 1), 4)}
if 'FETCH_SIZE' in g and 'WRITE_SIZE' in g:
    # KB per dispatch; FETCH_SIZE is doubled (the counter counts 64 B requests as 32 B on gfx950)
    derived["hbm_bytes_per_launch"] = int((2 * g['FETCH_SIZE'] + g['WRITE_SIZE']) * 1024)
    derived["fetch_kb_raw"], derived["write_kb"] = g['FETCH_SIZE'], g['WRITE_SIZE']
lines.append('derived ' + json.dumps(derived))
open(f'{OUT}/pmc_summary.txt', 'w').write('\n'.join(lines) + '\n')
print(open(f'{OUT}/kernel_stats_summary.txt').read())
print('\n'.join(lines))
PY
