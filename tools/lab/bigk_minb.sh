#!/bin/bash
# the column pass compiled for 2 / 3 / 4 workgroups per CU (-DBK_PASS_MINB), built and timed on the GPU box: tools/lab/bigk_minb.sh 3 2 4
. tools/lab/labbuild.sh
for n in "$@"; do
  lab_build gx_swav.hip -DBK_PASS_MINB=$n || exit 1
  echo "== BK_PASS_MINB=$n"
  python3 tools/lab/sinkhorn_rows_lab.py 2>&1 | grep "sorted    side by side True"
done
# ... and the pixel-major CE at 5000 columns compiled for one workgroup per CU (-DCE_PIX_NJ2=64) against two (80):
for n in 80 64; do
  lab_build gx_swav.hip -DCE_PIX_NJ2=$n || exit 1
  echo "== CE_PIX_NJ2=$n"
  python3 bench.py --prototypes 5000 --steps 3 --warmup 1 --no-cpu-baseline --reference-faithful off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['ms_per_step_by_kernel']
print('  step %.1f ms  ce %.1f' % (d['ms_per_step'], k['ce_fused_bigk_kernel']))"
done
