#!/bin/bash
# GPU tests, then an A/B of two builds of the library in one session (PTX355_LIB).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4 || exit 1
for i in 1 2; do
for lib in pathtracer_amd/lib/libptx355_prev.so pathtracer_amd/lib/libptx355.so; do
  echo "== $lib"
  PTX355_LIB=$lib PROBE_N=8,16 timeout -k 10 300 python3 tools/perf_probe.py all_mesh teapot 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
  PTX355_LIB=$lib PROBE_SIZE=1920x1080 PROBE_N=1,2,4 timeout -k 10 300 python3 tools/perf_probe.py cornell_box rings test aelita earth 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
  echo "-- back to back"
  PROBE_REPEAT=64 PTX355_LIB=$lib PROBE_SIZE=1920x1080 PROBE_N=1,2,4 timeout -k 10 300 python3 tools/perf_probe.py cornell_box rings test aelita earth 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //; s/bounces.path.*ptx/ptx/' || exit 1
done; done
