#!/bin/bash
# gx_gemm.hip rebuilt with the given compile flags and timed on the GPU box (tools/lab/sc_lab.py): tools/lab/gemm_flags.sh "" -DXW_ABL_DIRECT ...
root=$(pwd); pkg=$root/ganecdotes_amd
ls $pkg/build/*.o > /dev/null || exit 1
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $flags -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_gemm.hip -o $pkg/build/gx_gemm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== gx_gemm.hip with '$flags'"; python3 tools/lab/sc_lab.py 2>&1 | tail -2
done
