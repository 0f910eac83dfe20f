#!/bin/bash
# Short bench of every workload (GPU box).  Output: one summary line per workload.
set -e
for w in "cornell --steps 3" "teapot --steps 3" "fractals --steps 2" "all_mesh --mesh-quads 8 --steps 2"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:30], d['value'], 'Mpaths/s', d['gpath_bounce_per_s'], 'Gpb/s', d['ms_per_step'], 'ms', d['roofline']['kernel'])
"
done
