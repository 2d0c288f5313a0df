#!/bin/bash
# same-box A/B of environment knobs on the current library: tools/ab_env.sh <tag> "<VAR=a>" "<VAR=b>" ... [ROUNDS=n as first arg after tag]
TAG=$1; shift
ROUNDS=1
if [[ "$1" == ROUNDS=* ]]; then ROUNDS=${1#ROUNDS=}; shift; fi
mkdir -p gpurun_out
for r in $(seq 1 $ROUNDS); do
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 300 python bench.py --steps 8 --warmup 2 --cpu-baseline-n 0 --knm-iters 5 --kernel-pass-steps 4 > gpurun_out/${TAG}_${i}_$r.log 2>&1 || { echo "$e failed"; tail -5 gpurun_out/${TAG}_${i}_$r.log; continue; }
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/${TAG}_${i}_$r.log") if x.startswith("{")][-1])
print("%-24s round $r %.2f ms/step |" % ("$e", d["ms_per_step"]), " ".join("%s %.2f" % (a, b) for a, b in d["kernel_ms_per_step"].items()))
PY
done
done
