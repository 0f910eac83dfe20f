#!/bin/bash
# Every scene of the reference at 1080p: kernel and rate at 1, 4, 16 and 64 samples per launch, one launch at a time.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
S=$(ls data/scenes/*.npz | xargs -n1 basename | sed 's/.npz//' | tr '\n' ' ')
PROBE_SIZE=1920x1080 PROBE_N=1,4,16,64 timeout -k 10 1000 python3 tools/perf_probe.py $S 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/kernel //'
