#!/bin/bash
# like ab_env.sh, but prints every timing slot: tools/ab_full.sh <tag> "ENV=.." ...
TAG=$1; shift
mkdir -p gpurun_out
for E in "$@"; do
  NAME=$(echo "$E" | tr ' =' '__'); if [ "$E" = "-" ]; then E=""; NAME=base; fi
  env $E timeout -k 10 300 python bench.py --steps 8 --warmup 2 --cpu-baseline-n 0 --knm-iters 5 --kernel-pass-steps 4 > gpurun_out/${TAG}_$NAME.log 2>&1 || { echo "$NAME failed"; tail -5 gpurun_out/${TAG}_$NAME.log; exit 1; }
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/${TAG}_$NAME.log") if x.startswith("{")][-1])
print("%-24s %.2f ms/step |" % ("$NAME", d["ms_per_step"]), " ".join("%s %.2f" % (a, b) for a, b in d["kernel_ms_per_step"].items()))
PY
done
