#!/bin/bash
# Lab builds never touch the product library.  Source this file, then
#     lab_build gx_swav.hip -DCE_PIX_ABL=1
# compiles that ONE translation unit with -DGX_LAB and the given flags into ganecdotes_amd/build/lab/, links it with the
# production objects of every other translation unit into ganecdotes_amd/libganecdotes_lab.so and exports GX_LAB_LIB, which
# ganecdotes_amd/_lib.py honours (with a line on stderr).  libganecdotes_hip.so, its objects and its .srchash stamp stay as
# build.py made them, so a bench / pytest run after a lab session uses the product kernels.
lab_root=$(pwd); lab_pkg=$lab_root/ganecdotes_amd
lab_build() {
  local src=$1; shift
  local stem=${src%.hip}
  python3 -m ganecdotes_amd.build > /dev/null || return 1          # the production objects (no-op when the stamp matches)
  mkdir -p $lab_pkg/build/lab
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -Wno-unused-value -DGX_LAB "$@" \
      -I $lab_pkg/csrc -I $lab_root/include -c $lab_pkg/csrc/$src -o $lab_pkg/build/lab/$stem.o || return 1
  local objs=""
  for o in $lab_pkg/build/*.o; do
    [ "$(basename $o)" = "$stem.o" ] || objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lab_pkg/libganecdotes_lab.so $objs $lab_pkg/build/lab/$stem.o || return 1
  export GX_LAB_LIB=$lab_pkg/libganecdotes_lab.so
}
