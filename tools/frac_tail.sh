#!/bin/bash
# One-sample launches of fractals.yaml in turn (one in flight): how the launch's length depends on the jobs a wave reserves at a time.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for O in "flights=1" "flights=1,claim_tile=1" "flights=1,claim_tile=16" "flights=1,claim_tile=64" "flights=1,claim_tile=256" "host_waits_per_frame=0" "host_waits_per_frame=0,claim_tile=16"; do
  echo "== $O"
  PROBE_REPEAT=6 PROBE_OPTS=$O PROBE_SIZE=1920x1080 PROBE_N=1,2 timeout -k 10 600 python3 tools/perf_probe.py fractals 2>&1 | grep "n=\|rror" | sed 's/ Gpath-bounce.s//; s/bounces.path.*ptx/ptx/'
done
