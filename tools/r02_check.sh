#!/bin/bash
# GPU tests, then the probe lines of the mesh workloads (after a change to the scheduled kernels).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputest_check.log 2>&1 || { echo "gpu tests failed"; tail -20 gpurun_out/gputest_check.log; exit 1; }
tail -1 gpurun_out/gputest_check.log
for i in 1 2; do
PROBE_N=1,8 timeout -k 10 300 python3 tools/perf_probe.py all_mesh 2>/dev/null | grep "n=" | cut -c1-90
PROBE_N=1,16 timeout -k 10 300 python3 tools/perf_probe.py all 2>/dev/null | grep "n=" | cut -c1-90
done
