#!/bin/bash
# the blur pass at different register budgets (-DBLUR_MINB=n: workgroups per CU), built and timed on the GPU box:
#   tools/lab/blur_lb.sh 1 5 6
. tools/lab/labbuild.sh
for n in "$@"; do
  lab_build gx_conv.hip -DBLUR_MINB=$n || exit 1
  echo "== BLUR_MINB=$n"; python3 tools/lab/upconv_parts.py 2>&1 | grep "blur"
done
