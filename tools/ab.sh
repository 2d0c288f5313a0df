#!/bin/bash
# A/B on ONE GPU box (the pool's boxes differ by ~5 %): bench with the in-tree library, then with tools/<alt>.so copied over it, twice.
# Build the alternative first, e.g. an older commit's api.hip or a -D switch:
#   (cd gdrf_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGDRF_BWDKNM_SCALAR_EPILOGUE -shared -o ../../tools/alt.so api.hip)
#   gpurun -- 'bash tools/ab.sh alt.so'
set -e
ALT=$1; shift
L=gdrf_amd/csrc/libgdrf_hip.so
cp $L /tmp/lib_new.so
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-n 0 --knm-iters 2 "$@" 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('%.2f ms/step' % d['ms_per_step'], {n: round(v,2) for n,v in k.items() if v>1})"; }
echo NEW; run "$@"
cp tools/$ALT $L; echo ALT; run "$@"
cp /tmp/lib_new.so $L; echo NEW; run "$@"
cp tools/$ALT $L; echo ALT; run "$@"
cp /tmp/lib_new.so $L
