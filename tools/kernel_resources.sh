#!/bin/bash
# Register / scratch / occupancy report of every kernel in ptx355.hip (cross-compiles; no GPU needed).
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c pathtracer_amd/csrc/ptx355.hip -o /tmp/ptx_res.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - - 
