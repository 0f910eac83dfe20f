#!/bin/bash
# The per-scene job hand-out at three frame sizes: the defaults against one counter with fixed reservations of 64 (128 at 4K) jobs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for size in 1920x1080 3840x2160; do
for o in "claim_ranges=1,claim_paced=0" "claim_ranges=0"; do
  echo "== $size in turn [$o]"
  PROBE_OPTS="$o" PROBE_SIZE=$size PROBE_N=1,4 timeout -k 10 300 python3 tools/perf_probe.py test night aelita cube earth demo 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
  echo "== $size back to back [$o]"
  PROBE_REPEAT=32 PROBE_OPTS="$o" PROBE_SIZE=$size PROBE_N=1,4 timeout -k 10 300 python3 tools/perf_probe.py test night aelita cube earth demo 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
done; done
