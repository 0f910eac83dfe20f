#!/bin/bash
# Short launches back to back on meshes and the Moebius scenes, three frame sizes: the default (pools) against the wave-scheduled kernels (pool_over 4).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for S in 1280x720 1920x1080 3840x2160; do
for O in "host_waits_per_frame=0" "host_waits_per_frame=0,pool_over=4"; do
  echo "== $S $O"
  PROBE_REPEAT=64 PROBE_OPTS=$O PROBE_SIZE=$S PROBE_N=1,2,4 timeout -k 10 600 python3 tools/perf_probe.py teapot man car all mebius 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/'
done; done
