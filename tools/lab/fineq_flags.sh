#!/bin/bash
# gx_fineq.hip rebuilt with the given compile flags and timed on the GPU box, one build per argument: tools/lab/fineq_flags.sh "" -DFQ_LATE=0 ...
root=$(pwd); pkg=$root/ganecdotes_amd
ls $pkg/build/*.o > /dev/null || exit 1
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -Wno-unused-value $flags -I $pkg/csrc -I $root/include -c $pkg/csrc/gx_fineq.hip -o $pkg/build/gx_fineq.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libganecdotes_hip.so $pkg/build/*.o || exit 1
  echo "== gx_fineq.hip with '$flags'"; python3 tools/lab/fineq_lab.py fwd bwd 2>&1 | grep "fineq_fwd\|one pass"
done
