#!/bin/bash
# A/B of two builds of the library on the probe workloads, alternating, in one GPU-box call.
# usage: tools/ab_probe.sh <libA> <libB> ; PROBE cases through AB_CASES="teapot:1,16 all:16 all_mesh:1,8"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2
CASES=${AB_CASES:-teapot:1,16 all:1,16 all_mesh:1,8 cornell_box:1,128 fractals:1,32}
for c in $CASES; do
  name=${c%%:*}; ns=${c#*:}
  for L in $A $B $A $B; do
    echo "== $L"
    PTX355_LIB=$L PROBE_N=$ns timeout -k 10 200 python3 tools/perf_probe.py $name 2>&1 | grep "n=" || exit 1
  done
done
