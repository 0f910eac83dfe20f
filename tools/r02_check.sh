#!/bin/bash
# GPU tests after a kernel change, then the default bench line twice.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputest_check.log 2>&1 || { echo "gpu tests failed"; tail -30 gpurun_out/gputest_check.log; exit 1; }
tail -1 gpurun_out/gputest_check.log
for i in 1 2; do timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'Mpaths/s', d['ms_per_step'], 'ms/step', d['gpath_bounce_per_s'], d['roofline']['kernel'])" || exit 1; done
SORTED_N=16,32,64 timeout -k 10 200 python3 tools/sorted_probe.py 2>/dev/null | grep "sorted_pool=1" | tail -3
