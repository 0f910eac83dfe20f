#!/bin/bash
# The device library built with other code-generation flags (pathtracer_amd/lib/diag/libptx355_<name>.so) against the shipped build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in "" ${FLAG_LIBS:-ifcvt maxilp phifold both}; do
  if [ -n "$L" ]; then export PTX355_LIB=pathtracer_amd/lib/diag/libptx355_$L.so; fi
  echo "== ${L:-shipped}"
  PROBE_REPEAT=3 PROBE_SIZE=1920x1080 PROBE_N=32 timeout -k 10 300 python3 tools/perf_probe.py cornell_box all unit_logo 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/bounces.path.*ptx/ptx/'
  PROBE_REPEAT=2 PROBE_SIZE=1920x1080 PROBE_N=16 timeout -k 10 300 python3 tools/perf_probe.py fractals 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/bounces.path.*ptx/ptx/'
  PROBE_REPEAT=3 PROBE_SIZE=3840x2160 PROBE_N=16 timeout -k 10 300 python3 tools/perf_probe.py teapot 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/bounces.path.*ptx/ptx/'
done
