#!/bin/bash
# cache-policy bits on the many-prototype row kernels' loads / stores (-DBK_LOAD_AUX / -DBK_STORE_AUX), built and timed on the GPU box:
#   tools/lab/bigk_aux.sh "0 0" "2 0" "0 2" "2 2"
. tools/lab/labbuild.sh
for pair in "$@"; do
  set -- $pair
  lab_build gx_swav.hip -DBK_LOAD_AUX=$1 -DBK_STORE_AUX=$2 || exit 1
  echo "== load aux $1, store aux $2"
  python3 tools/lab/sinkhorn_rows_lab.py 2>&1 | grep "sorted    side by side True"
  python3 bench.py --prototypes 5000 --steps 3 --warmup 1 --no-cpu-baseline --reference-faithful off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['ms_per_step_by_kernel']
print('  step %.1f ms  sinkhorn %.1f  ce %.1f' % (d['ms_per_step'], k['sinkhorn_pass_kernel'], k['ce_fused_bigk_kernel']))"
done
