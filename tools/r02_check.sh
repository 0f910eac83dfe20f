#!/bin/bash
# GPU tests, then the pool-size sweep of the sorted-pool kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputest_check.log 2>&1 || { echo "gpu tests failed"; tail -30 gpurun_out/gputest_check.log; exit 1; }
tail -1 gpurun_out/gputest_check.log
for m in 1024 2048 4096 8192 16384; do
  echo "== pool_paths $m"
  PROBE_OPTS=pool_paths=$m SORTED_N=16,128 timeout -k 10 200 python3 tools/sorted_probe.py 2>/dev/null | grep "sorted_pool=1" | tail -2 || exit 1
done
