#!/bin/bash
# Launches in turn (flights=1) against overlapping launches (default), bench lines on the GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for W in cornell all_mesh fractals teapot; do
  for O in "" "--opt flights=1"; do
    echo "== $W $O"
    timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline --steps ${STEPS:-6} $O 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step  median', d['ms_per_step_median'], ' kernel_avg', r['kernel_avg_ms'], 'overlap', d['launches_overlap'])" || exit 1
  done
done
