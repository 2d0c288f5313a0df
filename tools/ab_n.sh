#!/bin/bash
# per-rank sizes of the 2/4/8-GPU runs on ONE GPU (no collective): tools/ab_n.sh <tag> N1 N2 ...
TAG=$1; shift
for N in "$@"; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-baseline-n 0 --knm-iters 2 --kernel-pass-steps 5 --n $N > gpurun_out/${TAG}_$N.log 2>&1 || { echo "$N failed"; tail -3 gpurun_out/${TAG}_$N.log; exit 1; }
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/${TAG}_$N.log") if x.startswith("{")][-1])
k=d["kernel_ms_per_step"]
print("N=%-8d %.2f ms/step |" % ($N, d["ms_per_step"]), " ".join("%s %.2f" % (a, b) for a, b in k.items()))
PY
done
