#!/bin/bash
# compile-time ablations of fineq_fwd_kernel (FQ_ABL), built and timed on the GPU box: tools/lab/fineq_abl.sh 0 1 2 ...
root=$(pwd); pkg=$root/ganecdotes_amd
for abl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -DFQ_ABL=$abl -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_fineq.hip -o $pkg/build/gx_fineq.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== FQ_ABL=$abl"; python3 tools/lab/fineq_lab.py fwd bwd 2>&1 | grep "fineq_fwd\|one pass"
done
