#!/bin/bash
# Short launches (1, 2, 4 samples) of every scene that runs the lockstep LDS kernels, at 1080p, back to back (64 launches) and in
# turn: the default (sorted path pools with a pool set per flight) against lane-owned sample jobs (pool_over 4).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="${SCENES:-aelita box cornell_box cube demo disc earth ellipsoid hyperboloid night paraboloid planet playground pyramid rings test tetrahedron torus}"
for MODE in back_to_back in_turn; do
  R=""; [ $MODE = back_to_back ] && R="PROBE_REPEAT=64"
  for O in "host_waits_per_frame=0" "host_waits_per_frame=0,pool_over=4" "host_waits_per_frame=-1"; do
    echo "== $MODE $O"
    env $R PROBE_OPTS=$O PROBE_SIZE=1920x1080 PROBE_N=1,2,4 timeout -k 10 600 python3 tools/perf_probe.py $S 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/'
  done
done
