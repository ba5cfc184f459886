#!/bin/bash
# compile-time ablations of fineq_fwd_kernel (FQ_ABL), built and timed on the GPU box: tools/lab/fineq_abl.sh 0 1 2 ...
. tools/lab/labbuild.sh
for abl in "$@"; do
  lab_build gx_fineq.hip -DFQ_ABL=$abl || exit 1
  echo "== FQ_ABL=$abl"; python3 tools/lab/fineq_lab.py fwd bwd 2>&1 | grep "fineq_fwd\|one pass"
done
