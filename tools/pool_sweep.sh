#!/bin/bash
# Pools: guided top-ups, the finish pass and the transposed job order, Cornell box and rings.yaml at 1080p.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for S in 0 1; do for G in 0 2; do
  F=256
  echo "== long launches pool_spread=$S pool_guided=$G pool_finish=$F"
  PROBE_OPTS=pool_spread=$S,pool_guided=$G,pool_finish=$F PROBE_SIZE=1920x1080 PROBE_N=16,128 timeout -k 10 300 python3 tools/perf_probe.py cornell_box rings 2>&1 | grep "n=" | sed 's/bounces.path.*ptx/ptx/'
  for PP in 2048 4096; do
  echo "== short launches back to back, pools $PP x 4 flights, pool_spread=$S pool_guided=$G pool_finish=$F"
  PROBE_REPEAT=64 PROBE_OPTS=host_waits_per_frame=0,sorted_pool=1,pool_over=0,pool_paths=$PP,flights=4,pool_spread=$S,pool_guided=$G,pool_finish=$F PROBE_SIZE=1920x1080 PROBE_N=1,2,4 timeout -k 10 300 python3 tools/perf_probe.py cornell_box rings 2>&1 | grep "n=" | sed 's/bounces.path.*ptx/ptx/'
  done
done; done
echo "== lane-owned, back to back"
PROBE_REPEAT=64 PROBE_OPTS=host_waits_per_frame=0 PROBE_SIZE=1920x1080 PROBE_N=1,2,4 timeout -k 10 300 python3 tools/perf_probe.py cornell_box rings 2>&1 | grep "n=" | sed 's/bounces.path.*ptx/ptx/'
