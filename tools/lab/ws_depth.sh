#!/bin/bash
# k-tiles in flight per producer thread of gemm_x6ws_kernel (-DWS_DEPTH=2: two register sets, the form of rounds 1-5; 4: four):
# the long-k TN GEMMs alone (dw_lab.py) and the steps.   usage: tools/lab/ws_depth.sh 2 4 2 4
. tools/lab/labbuild.sh
for n in "$@"; do
  lab_build gx_gemm.hip -DWS_DEPTH=$n || exit 1
  echo "== WS_DEPTH=$n"
  python3 bench.py --all-configs off --reference-faithful off --no-cpu-baseline --steps 40 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("K=8    loss", d["config"]["loss"], "step", round(d["value"],1), "images/s", round(d["ms_per_step"],3), "ms; one-stream gemm", d["roofline"]["one_stream_ms_per_step_by_kernel"]["gemm_f32_kernel"])'
  python3 bench.py --all-configs off --reference-faithful off --no-cpu-baseline --prototypes 5000 --steps 6 --warmup 2 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; k=r.get("one_stream_ms_per_step_by_kernel") or r["ms_per_step_by_kernel"]; print("K=5000 loss", d["config"]["loss"], "step", round(d["value"],2), "images/s", round(d["ms_per_step"],2), "ms; gemm", k["gemm_f32_kernel"])'
done
