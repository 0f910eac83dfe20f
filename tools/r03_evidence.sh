#!/bin/bash
# The measurement set kept under profiles/ (bench lines, kernel statistics, counter passes, pool phase cycles), on a GPU box.  usage: tools/r03_evidence.sh lines | profiles | phase
#   lines     the default bench line (all four workloads, tolerance modes of the fractal workload) and the one-sample-per-launch lines
#   profiles  rocprofv3 kernel trace + PMC passes of the four workloads (tools/profile_r02.sh; fractals with launches in turn)
#   phase     cycles per phase of the Cornell box's pool kernel (diagnostic build -DPTX_PHASE_TIMING in pathtracer_amd/lib/diag/)
# Outputs under gpurun_out/r3; copy what is to be kept into profiles/r03_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3; mkdir -p $O
PART=${1:-lines}
if [ "$PART" = lines ]; then
  timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
  tail -c 300 $O/bench.json; echo
  for W in cornell teapot fractals all_mesh; do
    timeout -k 10 300 python3 bench.py --workload $W --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_${W}_spp1.json 2> $O/bench_${W}_spp1.err || { echo "bench $W spp1 failed"; tail -5 $O/bench_${W}_spp1.err; exit 1; }
    tail -c 200 $O/bench_${W}_spp1.json; echo
  done
elif [ "$PART" = profiles ]; then
  for W in cornell teapot all_mesh; do
    bash tools/profile_r02.sh $W --no-other-workloads > $O/prof_$W.txt 2>&1 || { echo "$W profile failed"; tail -5 $O/prof_$W.txt; exit 1; }
    tail -1 $O/prof_$W.txt
  done
  bash tools/profile_r02.sh fractals --opt flights=1 > $O/prof_fractals.txt 2>&1 || { echo "fractals profile failed"; tail -5 $O/prof_fractals.txt; exit 1; }
  tail -1 $O/prof_fractals.txt
  # the reference's own dispatch: one-sample launches of the Cornell box back to back (kernel trace only: the launches overlap)
  export TMPDIR=/tmp
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_spp1 -o trace -- python3 bench.py --workload cornell --spp-per-step 1 --steps 64 --warmup 4 --no-cpu-baseline > $O/trace_spp1.log 2>&1 || { echo "spp1 trace failed"; tail -5 $O/trace_spp1.log; exit 1; }
  grep -h "^\"ptx_\|^\"Name" $(find $O/trace_spp1 -name "*kernel_stats.csv") > $O/kernel_stats_cornell_spp1.txt; cat $O/kernel_stats_cornell_spp1.txt
else
  PTX355_LIB=pathtracer_amd/lib/diag/libptx355_phase.so PROBE_N=1,16,128 timeout -k 10 200 python3 tools/perf_probe.py cornell_box > $O/phase_cornell.txt 2>&1 || { echo "phase probe failed"; exit 1; }
  cat $O/phase_cornell.txt
fi
