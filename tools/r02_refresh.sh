#!/bin/bash
# Refreshes one workload's evidence after a kernel change: bench lines (batched, one sample per launch) and rocprofv3 passes.
# usage: tools/r02_teapot.sh <workload>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-teapot}
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 400 python3 bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err || { echo "bench $W failed"; exit 1; }
timeout -k 10 300 python3 bench.py --workload $W --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_${W}_spp1.json 2> $O/bench_${W}_spp1.err || exit 1
bash tools/profile_r02.sh $W > $O/prof_$W.txt 2>&1 || { echo "$W profile failed"; exit 1; }
tail -1 $O/prof_$W.txt
