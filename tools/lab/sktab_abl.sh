#!/bin/bash
# the table Sinkhorn's sweeps with their arithmetic removed (-DSKT_ABL=1: the rows are loaded and dropped), built and timed on the box:
#   tools/lab/sktab_abl.sh 0 1
root=$(pwd); pkg=$root/ganecdotes_amd
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -DSKT_ABL=$v -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_sktab.hip -o $pkg/build/gx_sktab.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== SKT_ABL $v"
  python3 tools/lab/sktab_lab.py 5000 5 16 mask 2>&1 | grep "table sweep\|shifted"
done
