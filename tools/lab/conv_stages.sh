#!/bin/bash
# Ring depth of the f16x3 conv main loop (XH_STAGES: 16 KB of LDS per stage), layer by layer at the bench batch — each depth a
# LAB build (libganecdotes_lab.so), the product library untouched.   usage: bash tools/lab/conv_stages.sh [views]
. tools/lab/labbuild.sh
for s in 2 3 4; do
  lab_build gx_conv.hip -DXH_STAGES=$s || exit 1
  echo "== XH_STAGES=$s"
  python3 tools/lab/conv_lab.py ${1:-32} 2>&1 | grep -v "amdgpu\|intra-op\|LAB build"
done
