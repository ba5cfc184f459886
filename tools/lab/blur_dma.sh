#!/bin/bash
# the LDS-DMA blur pass at other prefetch depths / strip heights, built (lab library) and timed on the GPU box:
#   tools/lab/blur_dma.sh "-DBD_AHEAD=5 -DBD_ROWS=32" "-DBD_AHEAD=8 -DBD_ROWS=32" "-DBD_AHEAD=5 -DBD_ROWS=64"
. tools/lab/labbuild.sh
for flags in "$@"; do
  lab_build gx_conv.hip $flags || exit 1
  echo "== gx_conv.hip with '$flags'"; python3 tools/lab/upconv_parts.py 2>&1 | grep "blur"
done
