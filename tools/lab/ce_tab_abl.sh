#!/bin/bash
# build variants of the pixel-major CE (gx_bigk.h ce_pixels_bigk_kernel) timed on the GPU box: tools/lab/ce_tab_abl.sh K "flags" ...
# (-DCE_PIX_ABL=1 no scaling-vector loads, 2 no d(scores) stores, 3 a multiply for every exponential)
. tools/lab/labbuild.sh; K=$1; shift
for flags in "$@"; do
  lab_build gx_swav.hip $flags || exit 1
  echo "== '$flags'"; python3 tools/lab/ce_tab_lab.py $K 2>&1 | tail -2
done
