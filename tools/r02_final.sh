#!/bin/bash
# The round's closing measurement set, in two GPU-box calls (each within the box's time limit):
#   tools/r02_final.sh lines     bench lines of the four workloads (batched launches and one sample per launch), site statistics
#                                (diagnostic build -DPTX_SITE_STATS), phase cycles of the lockstep kernel (-DPTX_PHASE_TIMING)
#   tools/r02_final.sh profiles  rocprofv3 kernel trace + PMC passes of the four workloads (tools/profile_r02.sh)
# Outputs under gpurun_out/r2f; the diagnostic libraries are built beforehand into pathtracer_amd/lib/diag/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2f; mkdir -p $O
PART=${1:-lines}
if [ "$PART" = lines ]; then
  for W in cornell teapot fractals all_mesh; do
    timeout -k 10 400 python3 bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err || { echo "bench $W failed"; tail -5 $O/bench_$W.err; exit 1; }
    tail -c 600 $O/bench_$W.json; echo
  done
  for W in cornell teapot fractals all_mesh; do
    timeout -k 10 300 python3 bench.py --workload $W --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_${W}_spp1.json 2> $O/bench_${W}_spp1.err || { echo "bench $W spp1 failed"; tail -5 $O/bench_${W}_spp1.err; exit 1; }
  done
  S=pathtracer_amd/lib/diag/libptx355_sites.so
  PTX355_LIB=$S timeout -k 10 200 python3 tools/site_stats.py cornell_box 1920 1080 8 > $O/sites_cornell.txt 2>&1 || { echo "cornell site stats failed"; tail -5 $O/sites_cornell.txt; exit 1; }
  PTX355_LIB=$S timeout -k 10 300 python3 tools/site_stats.py fractals 1920 1080 8 > $O/sites_fractals.txt 2>&1 || { echo "fractal site stats failed"; tail -5 $O/sites_fractals.txt; exit 1; }
  PTX355_LIB=$S timeout -k 10 300 python3 tools/site_stats.py all 1920 1080 8 2236 > $O/sites_all_mesh.txt 2>&1 || { echo "mesh site stats failed"; tail -5 $O/sites_all_mesh.txt; exit 1; }
  PTX355_LIB=pathtracer_amd/lib/diag/libptx355_phase.so PROBE_N=1,128 timeout -k 10 120 python3 tools/perf_probe.py cornell_box > $O/phase_cornell.txt 2>&1 || { echo "phase probe failed"; exit 1; }
else
  for W in cornell teapot fractals all_mesh; do
    bash tools/profile_r02.sh $W > $O/prof_$W.txt 2>&1 || { echo "$W profile failed"; tail -5 $O/prof_$W.txt; exit 1; }
    tail -1 $O/prof_$W.txt
  done
fi
