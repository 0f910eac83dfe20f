#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_pools.py tests/test_gpu_scenes.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -8 || exit 1
for o in "sorted_pool=0" "sorted_pool=-1"; do
  echo "== [$o]"
  PROBE_OPTS="$o" PROBE_SIZE=1920x1080 PROBE_N=4,8,16,64 timeout -k 10 300 python3 tools/perf_probe.py unit_logo 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
done
