#!/bin/bash
# all.yaml + 10 M triangles: the share of live lanes that calls for a marcher / a walk to be served, swept (bench.py --opt).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for O in "" "--opt share_march=2" "--opt share_march=4" "--opt share_march=8" "--opt share_march=32" "--opt share_march=64" "--opt share_walk=2" "--opt share_walk=8" "--opt share_march=8 --opt share_walk=8"; do
  echo -n "$O: "
  timeout -k 10 120 python3 bench.py --workload all_mesh --steps 3 --warmup 1 --no-cpu-baseline $O 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(b['value'], b['ms_per_step'])"
done
