#!/bin/bash
# gx_gemm.hip rebuilt with the given compile flags and timed on the GPU box (tools/lab/sc_lab.py): tools/lab/gemm_flags.sh "" -DXW_ABL_DIRECT ...
. tools/lab/labbuild.sh
for flags in "$@"; do
  lab_build gx_gemm.hip $flags || exit 1
  echo "== gx_gemm.hip with '$flags'"; python3 tools/lab/sc_lab.py 2>&1 | tail -2
done
