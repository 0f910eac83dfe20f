#!/bin/bash
# Several builds of the library on the probe workloads, in turn, in one GPU-box call.
# usage: AB_CASES="teapot:1,16 all_mesh:1,8" tools/ab_multi.sh <lib> <lib> ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CASES=${AB_CASES:-teapot:1,16 all_mesh:1,8}
for c in $CASES; do
  name=${c%%:*}; ns=${c#*:}
  for L in "$@"; do
    echo "== $L"
    PTX355_LIB=$L PROBE_N=$ns PROBE_OPTS=$AB_OPTS timeout -k 10 200 python3 tools/perf_probe.py $name 2>&1 | grep "n=" || exit 1
  done
done
