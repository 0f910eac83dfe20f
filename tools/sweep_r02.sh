#!/bin/bash
# Shares of the scheduled kernels after the change to sample jobs (GPU box): small scenes with a Moebius strip.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== all (no mesh) 16 spp"
timeout -k 10 300 python3 tools/sweep_leaf.py 0 16 4,2,4,4 2,2,4,4 2,2,8,4 1,2,8,4 2,2,16,4 3,2,8,4 2,1,8,4 2,3,8,4 1,2,4,4 1,1,8,4 2,2,8,4 4,2,4,4 || exit 1
echo "== all (no mesh) 1 spp"
timeout -k 10 300 python3 tools/sweep_leaf.py 0 1 4,2,4,4 2,2,4,4 2,2,8,4 1,2,8,4 || exit 1
echo "== mebius scene 16 spp"
SWEEP_SCENE=mebius timeout -k 10 300 python3 tools/sweep_leaf.py 0 16 4,2,4,4 2,2,4,4 2,2,8,4 1,2,8,4 || exit 1
echo "== rings (torus) forced scheduled? default family"
SWEEP_SCENE=rings timeout -k 10 300 python3 tools/sweep_leaf.py 0 16 4,2,4,4 2,2,8,4 || exit 1
