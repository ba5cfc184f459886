#!/bin/bash
# gx_fineq.hip rebuilt with the given compile flags and timed on the GPU box, one build per argument: tools/lab/fineq_flags.sh "" -DFQ_LATE=0 ...
. tools/lab/labbuild.sh
for flags in "$@"; do
  lab_build gx_fineq.hip $flags || exit 1
  echo "== gx_fineq.hip with '$flags'"; python3 tools/lab/fineq_lab.py fwd bwd 2>&1 | grep "fineq_fwd\|one pass"
done
