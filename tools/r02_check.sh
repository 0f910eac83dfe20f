#!/bin/bash
# Pool tests, then the A/B of the sorted-pool kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_pools.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputest_check.log 2>&1 || { echo "gpu tests failed"; tail -30 gpurun_out/gputest_check.log; exit 1; }
tail -1 gpurun_out/gputest_check.log
SORTED_N=16,128 timeout -k 10 200 python3 tools/sorted_probe.py 2>/dev/null | grep "sorted_pool=1" || exit 1
