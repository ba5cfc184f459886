#!/bin/bash
# compile-time ablations of the fused finish epilogue of gemm_xw_kernel (FIN_ABL), built and timed on the GPU box:
#   tools/lab/fin_abl.sh 0 1 2 3 5
root=$(pwd); pkg=$root/ganecdotes_amd
for abl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -DFIN_ABL=$abl -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_gemm.hip -o $pkg/build/gx_gemm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== FIN_ABL=$abl"; python3 tools/lab/fineq_lab.py bwd 2>&1 | grep "epilogue"
done
