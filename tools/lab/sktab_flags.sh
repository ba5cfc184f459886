#!/bin/bash
# gx_sktab.hip rebuilt with the given compile flags and timed on the GPU box: tools/lab/sktab_flags.sh K "flags" ...  (-DSKT_FORCE_WIDE: the wide form wherever it fits)
. tools/lab/labbuild.sh; K=$1; shift
for flags in "$@"; do
  lab_build gx_sktab.hip $flags || exit 1
  echo "== gx_sktab.hip with '$flags'"; python3 tools/lab/sktab_lab.py $K 5 16 mask 2>&1 | grep "table\|shifted"
done
