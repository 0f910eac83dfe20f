#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do for o in "claim_paced=0" "claim_paced=-1"; do echo -n "$o: "; timeout -k 10 300 python3 bench.py --workload cornell --spp-per-step 1 --steps 128 --warmup 8 --no-cpu-baseline --opt $o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1; done; done
