#!/bin/bash
# Shares of the fractal scenes (GPU box).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== fractals 1080p 32 spp"
SWEEP_SCENE=fractals timeout -k 10 600 python3 tools/sweep_leaf.py 0 32 64,8,4,4 64,4,4,4 64,16,4,4 16,8,4,4 8,8,4,4 64,8,2,4 64,8,8,4 64,8,16,4 32,8,4,4 64,12,4,4 64,8,4,4 2>/dev/null | grep shares || exit 1
for sc in julia mandelbulb; do
  echo "== $sc 1080p 16 spp"
  SWEEP_SCENE=$sc timeout -k 10 600 python3 tools/sweep_leaf.py 0 16 64,8,4,4 16,8,4,4 64,4,4,4 64,8,8,4 2>/dev/null | grep shares || exit 1
done
