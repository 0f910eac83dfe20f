#!/bin/bash
# One GPU-box call for the round's measurement set: mesh site statistics (deep-stack traffic), PMC profiles of the
# Cornell and mesh workloads, the fractal serving experiment.  Outputs under gpurun_out/r2 and gpurun_out/prof_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2
PTX355_LIB=pathtracer_amd/lib/diag/libptx355_sites.so timeout -k 10 300 python tools/site_stats.py all 1920 1080 8 2236 > gpurun_out/r2/sites_mesh.txt 2>&1 || { echo "site stats failed"; tail -5 gpurun_out/r2/sites_mesh.txt; exit 1; }
for L in pathtracer_amd/lib/libptx355.so pathtracer_amd/lib/diag/libptx355_glue.so; do
  [ -f $L ] && { echo "== $L"; PTX355_LIB=$L PROBE_N=32 timeout -k 10 200 python tools/perf_probe.py fractals 2>&1 | grep "n=" || exit 1; }
done
bash tools/profile_r02.sh cornell > gpurun_out/r2/prof_cornell.txt 2>&1 || { echo "cornell profile failed"; tail -5 gpurun_out/r2/prof_cornell.txt; exit 1; }
bash tools/profile_r02.sh all_mesh > gpurun_out/r2/prof_all_mesh.txt 2>&1 || { echo "mesh profile failed"; tail -5 gpurun_out/r2/prof_all_mesh.txt; exit 1; }
tail -2 gpurun_out/r2/prof_cornell.txt; tail -2 gpurun_out/r2/prof_all_mesh.txt; grep -E "kernel=|BVH records|stack" gpurun_out/r2/sites_mesh.txt
