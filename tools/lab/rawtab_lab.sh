#!/bin/bash
# gx_proto_raw_tables (Zc -> Zc Wc^T, |Zc|^2: upsample2_add_scores_kernel<., false, true>) at different grid sizes
# (-DRAWTAB_BLOCKS=n), built and timed on the GPU box:   tools/lab/rawtab_lab.sh 512 1024 2048 4096
. tools/lab/labbuild.sh
for n in "$@"; do
  lab_build gx_swav.hip -DRAWTAB_BLOCKS=$n || exit 1
  echo "== RAWTAB_BLOCKS=$n"; python3 tools/lab/rawtab_lab.py
done
