#!/bin/bash
# GPU tests after a kernel change, then probes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputest_check.log 2>&1 || { echo "gpu tests failed"; tail -30 gpurun_out/gputest_check.log; exit 1; }
tail -1 gpurun_out/gputest_check.log
PROBE_N=1,4,16 timeout -k 10 300 python3 tools/perf_probe.py all mebius 2>/dev/null | grep "n=" | cut -c1-100
