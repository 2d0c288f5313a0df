#!/bin/bash
# timing experiments: bench with alternative builds of the library (results of those builds are wrong by design)
L=gdrf_amd/csrc/libgdrf_hip.so
cp $L /tmp/lib_new.so
run() { timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-baseline-n 0 --knm-iters 1 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('%.2f ms/step' % d['ms_per_step'], {n: round(v,2) for n,v in k.items() if v>5})"; }
echo BASE; run
for a in "$@"; do cp tools/$a $L; echo $a; run; done
cp /tmp/lib_new.so $L
