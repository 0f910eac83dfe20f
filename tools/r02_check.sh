#!/bin/bash
# Pools against lane-owned paths on the reference's small scenes (all staged in LDS), 1080p, 16 samples per launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sc in demo night planet test earth playground disc cube pyramid rings julia; do
  a=$(PROBE_OPTS="sorted_pool=0" PROBE_N=16 timeout -k 10 300 python3 tools/perf_probe.py $sc 2>/dev/null | grep "n=" | sed 's/.*kernel \([0-9.]* ms\).*bounces.path \([0-9.]*\).*/\1 (\2 bounces per path)/') || exit 1
  b=$(PROBE_N=16 timeout -k 10 300 python3 tools/perf_probe.py $sc 2>/dev/null | grep "n=" | sed 's/.*kernel \([0-9.]* ms\).*/\1/') || exit 1
  echo "$sc: lane-owned $a, default $b"
done
