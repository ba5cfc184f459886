#!/bin/bash
# compile-time variants of gx_conv.hip built and timed on the GPU box, baseline first: tools/lab/conv_abl.sh "" -DXP_ABL_A2 ...
root=$(pwd); pkg=$root/ganecdotes_amd
ls $pkg/build/*.o > /dev/null || exit 1
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $flags -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_conv.hip -o $pkg/build/gx_conv.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== gx_conv.hip with '$flags'"; python3 tools/lab/conv_lab.py 2>&1 | grep "H=\|sum"
done
