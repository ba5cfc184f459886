#!/bin/bash
# build variants of the pixel-major CE (gx_bigk.h ce_pixels_bigk_kernel) timed on the GPU box: tools/lab/ce_tab_abl.sh K "flags" ...
# (-DCE_PIX_ABL=1 no scaling-vector loads, 2 no d(scores) stores, 3 a multiply for every exponential)
root=$(pwd); pkg=$root/ganecdotes_amd; K=$1; shift
ls $pkg/build/*.o > /dev/null || exit 1
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $flags -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_swav.hip -o $pkg/build/gx_swav.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== '$flags'"; python3 tools/lab/ce_tab_lab.py $K 2>&1 | tail -2
done
