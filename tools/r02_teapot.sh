#!/bin/bash
# After a change of the mesh shares: GPU tests, then the teapot's and all.yaml's bench lines and the teapot's rocprofv3 passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gputest.log 2>&1 || { echo "gpu tests failed"; tail -20 $O/gputest.log; exit 1; }
tail -1 $O/gputest.log
timeout -k 10 400 python3 bench.py --workload teapot > $O/bench_teapot.json 2> $O/bench_teapot.err || { echo "bench teapot failed"; exit 1; }
timeout -k 10 300 python3 bench.py --workload teapot --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_teapot_spp1.json 2> $O/bench_teapot_spp1.err || exit 1
bash tools/profile_r02.sh teapot > $O/prof_teapot.txt 2>&1 || { echo "teapot profile failed"; exit 1; }
tail -1 $O/prof_teapot.txt
PROBE_N=1,16 timeout -k 10 200 python3 tools/perf_probe.py all 2>/dev/null | grep "n="
