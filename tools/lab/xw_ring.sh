#!/bin/bash
# A ring depth of gemm_xw_kernel's f16x3 256-row form (-DXW_TALL_NS=n; past 2 the loop takes a raw barrier and a counted vmcnt):
# the pyramid / score shapes alone (sc_lab.py) and the step.   usage: tools/lab/xw_ring.sh 2 3 4
. tools/lab/labbuild.sh
for n in "$@"; do
  lab_build gx_gemm.hip -DXW_TALL_NS=$n || exit 1
  echo "== XW_TALL_NS=$n"
  python3 tools/lab/sc_lab.py 2>&1 | grep -v "amdgpu\|intra-op\|LAB build" | head -12
  python3 bench.py --all-configs off --reference-faithful off --no-cpu-baseline --steps 40 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("loss", d["config"]["loss"], "step", round(d["value"],1), "images/s", round(d["ms_per_step"],3), "ms; one-stream gemm", d["roofline"]["one_stream_ms_per_step_by_kernel"]["gemm_f32_kernel"])'
done
