#!/bin/bash
# the table Sinkhorn's sweeps with their arithmetic removed (-DSKT_ABL=1: the rows are loaded and dropped), built and timed on the box:
#   tools/lab/sktab_abl.sh 0 1
. tools/lab/labbuild.sh
for v in "$@"; do
  lab_build gx_sktab.hip -DSKT_ABL=$v || exit 1
  echo "== SKT_ABL $v"
  python3 tools/lab/sktab_lab.py 5000 5 16 mask 2>&1 | grep "table sweep\|shifted"
done
