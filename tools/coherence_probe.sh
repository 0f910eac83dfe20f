#!/bin/bash
# tools/coherence_probe.py under rocprofv3: per-dispatch durations of the query kernel for the four ray orders (GPU box).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
S=${1:-cornell_box}
OUT=gpurun_out/coh_$S; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 tools/coherence_probe.py $S > $OUT/log.txt 2>&1 || { echo "probe failed"; tail -5 $OUT/log.txt; exit 1; }
grep -v "^[WE]2026" $OUT/log.txt | grep -v amdgpu.ids
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('ptx_kat_trace')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for i, r in enumerate(rows):
    print(i, r['Kernel_Name'], 'grid', r.get('Grid_Size_X', r.get('Grid_Size', '?')), f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.3f} ms")
PY
