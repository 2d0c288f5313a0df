#!/bin/bash
# usage (through gpurun): tools/ab_env.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   - one short bench run per environment setting ("-" = none)
TAG=$1; shift
mkdir -p gpurun_out
for E in "$@"; do
  NAME=$(echo "$E" | tr ' =' '__')
  if [ "$E" = "-" ]; then E=""; NAME=base; fi
  env $E timeout -k 10 300 python bench.py --steps 8 --warmup 2 --cpu-baseline-n 0 --knm-iters 5 --kernel-pass-steps 4 > gpurun_out/${TAG}_$NAME.log 2>&1 || { echo "$NAME failed"; tail -5 gpurun_out/${TAG}_$NAME.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/${TAG}_$NAME.log") if x.startswith("{")][-1]
d=json.loads(l)
k=d["kernel_ms_per_step"]
print("%-28s %.2f ms/step | fwd_t %.2f wbar %.2f tn_sym %.2f tn_gt %.2f fwd_w %.2f bwd_knm %.2f knm %.2f rows %.2f | loss %.6f" % ("$NAME", d["ms_per_step"], k["fwd_t"], k["bwd_wbar"], k["tn_sym"], k["tn_gt"], k["fwd_w"], k["bwd_knm"], k["k_nm"], k["elbo_rows"], d["final_loss"]))
PY
done
