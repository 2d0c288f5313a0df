#!/bin/bash
# same-box A/B of prebuilt library variants: tools/ab_libs.sh <tag> <dir with *.so> [rounds]   (each variant replaces libgdrf_hip.so for its runs)
TAG=$1; DIR=$2; ROUNDS=${3:-2}
mkdir -p gpurun_out
cp gdrf_amd/csrc/libgdrf_hip.so /tmp/lib_orig.so
for r in $(seq 1 $ROUNDS); do
for f in $DIR/*.so; do
  NAME=$(basename $f .so)
  cp $f gdrf_amd/csrc/libgdrf_hip.so
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --cpu-baseline-n 0 --knm-iters 5 --kernel-pass-steps 4 > gpurun_out/${TAG}_${NAME}_$r.log 2>&1 || { echo "$NAME failed"; tail -5 gpurun_out/${TAG}_${NAME}_$r.log; continue; }
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/${TAG}_${NAME}_$r.log") if x.startswith("{")][-1])
print("%-8s round $r %.2f ms/step |" % ("$NAME", d["ms_per_step"]), " ".join("%s %.2f" % (a, b) for a, b in d["kernel_ms_per_step"].items()))
PY
done
done
cp /tmp/lib_orig.so gdrf_amd/csrc/libgdrf_hip.so
