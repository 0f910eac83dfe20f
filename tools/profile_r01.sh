#!/bin/bash
# Collects the round's rocprofv3 evidence for the benchmark command on a GPU box: kernel trace + stats,
# then one PMC pass per counter group (each its own run), and writes text summaries under
# gpurun_out/prof/ for copying into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
# kernel time of the other workloads (kernel trace only)
for w in teapot fractals; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o trace -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace_$w.log 2>&1 || { echo "trace pass $w failed"; tail -5 $OUT/trace_$w.log; exit 1; }
done
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_all_mesh -o trace -- python3 bench.py --workload all_mesh --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace_all_mesh.log 2>&1 || { echo "trace pass all_mesh failed"; tail -5 $OUT/trace_all_mesh.log; exit 1; }
run_pmc() { # name counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
run_pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU
run_pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - <<'PY'
import csv, glob, json, collections
out = {}
stats = sorted(glob.glob('gpurun_out/prof/trace*/**/*kernel_stats.csv', recursive=True))
rows = []
for f in stats:
    rows += [dict(Run=f.split('/')[2], **r) for r in csv.DictReader(open(f)) if r['Name'].startswith('ptx_')]
open('gpurun_out/prof/kernel_stats_summary.txt', 'w').write('\n'.join(','.join(f'{k}={v}' for k, v in r.items()) for r in rows) + '\n')
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/prof/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('ptx_'):
            a = agg[(r['Kernel_Name'], r['Counter_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
lines = [f'{k[0]},{k[1]},sum={v[0]:.0f},dispatches={v[1]},per_dispatch={v[0]/max(v[1],1):.1f}' for k, v in sorted(agg.items())]
open('gpurun_out/prof/pmc_summary.txt', 'w').write('\n'.join(lines) + '\n')
g = {k[1]: v[0] / max(v[1], 1) for k, v in agg.items()}
if 'FETCH_SIZE' in g and 'WRITE_SIZE' in g:
    # KB per dispatch; FETCH_SIZE is doubled (the counter counts 64 B requests as 32 B on gfx950)
    t = {"workload": "cornell", "spp_per_step": 128, "hbm_bytes_per_launch": int((2 * g['FETCH_SIZE'] + g['WRITE_SIZE']) * 1024),
         "fetch_size_kb_per_launch_raw": g['FETCH_SIZE'], "write_size_kb_per_launch": g['WRITE_SIZE'],
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline`; FETCH_SIZE doubled (gfx950 correction)"}
    if 'GRBM_GUI_ACTIVE' in g and 'SQ_ACTIVE_INST_VALU' in g:
        simd_cycles = g['GRBM_GUI_ACTIVE'] / 8 * 1024  # the counter sums 8 XCDs; 256 CUs x 4 SIMDs
        t["valu_issue"] = {"valu_busy_frac": round(g['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles, 4),
                           "lane_utilisation": round(g['SQ_THREAD_CYCLES_VALU'] / (g['SQ_ACTIVE_INST_VALU'] * 64), 4),
                           "valu_wave_instructions_per_launch": g['SQ_INSTS_VALU'],
                           "note": "SQ_ACTIVE_INST_VALU x 4 cycles over GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs; SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)"}
    json.dump(t, open('gpurun_out/prof/traffic.json', 'w'), indent=1)
print(open('gpurun_out/prof/kernel_stats_summary.txt').read())
print('\n'.join(lines))
PY
