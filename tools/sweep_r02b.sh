#!/bin/bash
# Mesh shares on the reference's other model scenes (GPU box).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sc in iron_man car man mill dead_tree unit_logo; do
  echo "== $sc 1080p 16 spp"
  SWEEP_SCENE=$sc timeout -k 10 300 python3 tools/sweep_leaf.py 0 16 64,2,4,4 8,2,4,4 4,2,4,4 2,2,4,4 64,2,4,4 4,2,4,4 2>/dev/null | grep shares || exit 1
done
