#!/bin/bash
# Sweeps the scheduling shares (walk,post,march) and the claim tile on one scene (GPU box): tools/sweep_shares.sh <scene> <n>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PROBE_N=${2:-8}
for tile in 1 64; do
  for sh in 8,4,4 16,4,4 64,4,4 16,8,4 64,8,4 64,16,4 64,64,4 64,8,2 64,8,8 64,64,64; do
    IFS=, read -r sw sp sm <<< "$sh"
    r=$(PROBE_OPTS="claim_tile=$tile,share_walk=$sw,share_post=$sp,share_march=$sm" timeout -k 10 100 python tools/perf_probe.py "$1" 2>/dev/null | grep "n=" | sed 's/.*kernel \([0-9.]* ms\).*/\1/' | tr '\n' ' ') || exit 1
    echo "tile $tile shares $sh: $r"
  done
done
