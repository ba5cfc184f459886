#!/bin/bash
# the blur pass at different register budgets (-DBLUR_MINB=n: workgroups per CU), built and timed on the GPU box:
#   tools/lab/blur_lb.sh 1 5 6
root=$(pwd); pkg=$root/ganecdotes_amd
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -DBLUR_MINB=$n -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_conv.hip -o $pkg/build/gx_conv.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== BLUR_MINB=$n"; python3 tools/lab/upconv_parts.py 2>&1 | grep "blur"
done
