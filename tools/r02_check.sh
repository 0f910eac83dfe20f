#!/bin/bash
# One sample per dispatch: lane-owned overlapping launches against pools in turn (experiment).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in "" "--opt sorted_pool=2" "--opt flights=1"; do
  echo "== cornell spp1 $o"
  timeout -k 10 300 python3 bench.py --workload cornell --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline $o 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel'])" || exit 1
done
for n in 2 4; do for o in "" "--opt sorted_pool=2"; do
  echo "== cornell spp$n $o"
  timeout -k 10 300 python3 bench.py --workload cornell --spp-per-step $n --steps 32 --warmup 4 --no-cpu-baseline $o 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel'])" || exit 1
done; done
