#!/bin/bash
# Closing evidence of the round for the four workloads: bench lines (batched, one sample per launch) and the rocprofv3 passes
# (the fractal workload profiled with launches in turn, so that a kernel's duration is a quantity of its own).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2f; mkdir -p $O
for W in cornell teapot all_mesh; do
  bash tools/r02_refresh.sh $W || exit 1
done
W=fractals
timeout -k 10 400 python3 bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err || { echo "bench $W failed"; exit 1; }
timeout -k 10 300 python3 bench.py --workload $W --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_${W}_spp1.json 2> $O/bench_${W}_spp1.err || exit 1
bash tools/profile_r02.sh fractals --opt flights=1 > $O/prof_fractals.txt 2>&1 || { echo "fractals profile failed"; tail -5 $O/prof_fractals.txt; exit 1; }
tail -1 $O/prof_fractals.txt
