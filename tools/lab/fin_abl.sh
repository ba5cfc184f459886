#!/bin/bash
# compile-time ablations of the fused finish epilogue of gemm_xw_kernel (FIN_ABL), built and timed on the GPU box:
#   tools/lab/fin_abl.sh 0 1 2 3 5
. tools/lab/labbuild.sh
for abl in "$@"; do
  lab_build gx_gemm.hip -DFIN_ABL=$abl || exit 1
  echo "== FIN_ABL=$abl"; python3 tools/lab/fineq_lab.py bwd 2>&1 | grep "epilogue"
done
