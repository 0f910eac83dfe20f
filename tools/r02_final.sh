#!/bin/bash
# One GPU-box call for the round's closing measurement set: bench lines of the four workloads (batched launches and
# one sample per launch), the fractal workload's rocprofv3 passes and site statistics.  Outputs under gpurun_out/r2f.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2f; mkdir -p $O
for W in cornell teapot fractals all_mesh; do
  timeout -k 10 400 python3 bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err || { echo "bench $W failed"; tail -5 $O/bench_$W.err; exit 1; }
  tail -c 600 $O/bench_$W.json; echo
done
for W in cornell teapot fractals all_mesh; do
  timeout -k 10 300 python3 bench.py --workload $W --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_${W}_spp1.json 2> $O/bench_${W}_spp1.err || { echo "bench $W spp1 failed"; tail -5 $O/bench_${W}_spp1.err; exit 1; }
done
PTX355_LIB=pathtracer_amd/lib/diag/libptx355_sites.so timeout -k 10 120 python3 tools/site_stats.py fractals 256 144 1 > $O/sites_fractals_small.txt 2>&1 || { echo "small site stats failed"; tail -5 $O/sites_fractals_small.txt; exit 1; }
PTX355_LIB=pathtracer_amd/lib/diag/libptx355_sites.so timeout -k 10 300 python3 tools/site_stats.py fractals 1920 1080 8 > $O/sites_fractals.txt 2>&1 || { echo "site stats failed"; tail -5 $O/sites_fractals.txt; exit 1; }
bash tools/profile_r02.sh fractals > $O/prof_fractals.txt 2>&1 || { echo "fractal profile failed"; tail -5 $O/prof_fractals.txt; exit 1; }
tail -3 $O/prof_fractals.txt
for W in cornell teapot all_mesh; do
  bash tools/profile_r02.sh $W > $O/prof_$W.txt 2>&1 || { echo "$W profile failed"; tail -5 $O/prof_$W.txt; exit 1; }
  tail -1 $O/prof_$W.txt
done
PTX355_LIB=pathtracer_amd/lib/diag/libptx355_phase.so PROBE_N=1,128 timeout -k 10 120 python3 tools/perf_probe.py cornell_box > $O/phase_cornell.txt 2>&1 || { echo "phase probe failed"; exit 1; }
