#!/bin/bash
# Timing-only variants of the wide sweep kernel (BHIP_DIAG in kernels_sweepw.hip): 1 = no x gather,
# 2 = no staged column indices (gathers its own row), 3 = no index loads at all (pure value stream in the
# sweep's own structure).  Results are WRONG by construction; only tools/probes/diag_sweep.py loads these.
set -e
cd "$(dirname "$0")/../../blasted_amd/csrc"
make -s
for d in ${DIAGS:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DBHIP_DIAG=$d -c kernels_sweepw.hip -o build/kernels_sweepw_diag$d.o
  objs=$(ls build/*.o | grep -v -e "kernels_sweepw\.o" -e "_diag")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libblasted_hip_diag$d.so $objs build/kernels_sweepw_diag$d.o
done
ls -la ../lib
