#!/bin/bash
# compile-time variants of gx_conv.hip built and timed on the GPU box, baseline first: tools/lab/conv_abl.sh "" -DXP_ABL_A2 ...
. tools/lab/labbuild.sh
for flags in "$@"; do
  lab_build gx_conv.hip $flags || exit 1
  echo "== gx_conv.hip with '$flags'"; python3 tools/lab/conv_lab.py 2>&1 | grep "H=\|sum"
done
