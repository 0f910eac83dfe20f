#!/bin/bash
# Short launches back to back on a lit mesh: the defaults against the pool knobs one at a time.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for O in "host_waits_per_frame=0" "host_waits_per_frame=0,pool_paths=2048" "host_waits_per_frame=0,pool_spread=0" "host_waits_per_frame=0,pool_paths=2048,pool_spread=0"; do
  echo "== $O"
  PROBE_REPEAT=32 PROBE_OPTS=$O PROBE_SIZE=1920x1080 PROBE_N=1,2 timeout -k 10 600 python3 tools/perf_probe.py unit_logo 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/'
done
