#!/bin/bash
# The per-scene job hand-out at three frame sizes: sixteen counters + paced reservations against one counter, fixed reservations.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_scenes.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
for size in 1280x720 1920x1080 3840x2160; do
for o in "claim_ranges=1,claim_paced=0" "claim_ranges=0"; do
  echo "== $size in turn [$o]"
  PROBE_OPTS="$o" PROBE_SIZE=$size PROBE_N=1,4 timeout -k 10 300 python3 tools/perf_probe.py test aelita cube earth 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
  echo "== $size back to back [$o]"
  PROBE_REPEAT=32 PROBE_OPTS="$o" PROBE_SIZE=$size PROBE_N=1,4 timeout -k 10 300 python3 tools/perf_probe.py test aelita cube earth 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
done; done
