#!/bin/bash
# Teapot 4K: sample jobs + fold against whole-pixel jobs per launch length, with the mesh shares 8/2/4 (GPU box).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for ss in 0 64; do
  echo "== teapot 4K split_small=$ss (0: whole-pixel jobs always; 64: sample jobs up to 64 samples per launch)"
  PROBE_OPTS="split_small=$ss" PROBE_N=1,2,4,8,16 timeout -k 10 300 python3 tools/perf_probe.py teapot 2>/dev/null | grep "n=" | cut -c1-80 || exit 1
done
