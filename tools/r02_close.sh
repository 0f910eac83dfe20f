#!/bin/bash
# GPU tests, then the rocprofv3 passes of the four workloads (the fractal workload with launches in turn, so that a kernel's
# duration is a quantity of its own).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gputest.log 2>&1 || { echo "gpu tests failed"; tail -20 $O/gputest.log; exit 1; }
tail -1 $O/gputest.log
for W in cornell teapot all_mesh; do
  bash tools/profile_r02.sh $W > $O/prof_$W.txt 2>&1 || { echo "$W profile failed"; tail -5 $O/prof_$W.txt; exit 1; }
  tail -1 $O/prof_$W.txt
done
bash tools/profile_r02.sh fractals --opt flights=1 > $O/prof_fractals.txt 2>&1 || { echo "fractals profile failed"; tail -5 $O/prof_fractals.txt; exit 1; }
tail -1 $O/prof_fractals.txt
