#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5 || exit 1
PROBE_SIZE=1920x1080 PROBE_N=1,2,4,16 timeout -k 10 300 python3 tools/perf_probe.py unit_logo 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/'
PROBE_REPEAT=32 PROBE_SIZE=1920x1080 PROBE_N=1,2,4 timeout -k 10 300 python3 tools/perf_probe.py unit_logo 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/'
