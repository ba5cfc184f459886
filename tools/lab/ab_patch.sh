#!/bin/bash
# A B A B on one box:   tools/lab/ab_patch.sh fqb [bench arguments]   ->  value / ms_per_step of each run
p=$1; shift
for v in none $p none $p; do
  echo -n "$v: "
  python3 tools/lab/ab_patch.py $v -- --all-configs off --reference-faithful off --no-cpu-baseline "$@" 2>/dev/null \
    | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "images/s", round(d["ms_per_step"],3), "ms;", {k: v for k, v in d["roofline"]["one_stream_ms_per_step_by_kernel"].items() if k in ("fineq_bwd_kernel", "gemm_f32_kernel", "phase:pyramid_bwd")})'
done
