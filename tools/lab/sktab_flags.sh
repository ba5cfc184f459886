#!/bin/bash
# gx_sktab.hip rebuilt with the given compile flags and timed on the GPU box: tools/lab/sktab_flags.sh K "flags" ...  (-DSKT_FORCE_WIDE: the wide form wherever it fits)
root=$(pwd); pkg=$root/ganecdotes_amd; K=$1; shift
ls $pkg/build/*.o > /dev/null || exit 1
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $flags -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_sktab.hip -o $pkg/build/gx_sktab.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== gx_sktab.hip with '$flags'"; python3 tools/lab/sktab_lab.py $K 5 16 mask 2>&1 | grep "table\|shifted"
done
