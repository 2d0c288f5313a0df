#!/bin/bash
# usage (through gpurun, from the repo root): tools/gpu_round.sh <tag> "<pytest args>" [bench args...]
# Runs pytest (all output to gpurun_out/<tag>_pytest.log), then - unless pytest hung - two short bench runs.
TAG=$1; shift; PT=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest $PT -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1
RC=$?
tail -15 gpurun_out/${TAG}_pytest.log
if [ $RC -eq 124 ] || [ $RC -eq 137 ]; then echo "pytest timed out: no further GPU step"; exit $RC; fi
for MODE in "$@"; do
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --cpu-baseline-n 0 --knm-iters 10 --mfma-mode $MODE > gpurun_out/${TAG}_bench_$MODE.log 2>&1 || { echo "bench $MODE failed"; tail -5 gpurun_out/${TAG}_bench_$MODE.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/${TAG}_bench_$MODE.log") if x.startswith("{")][-1]
d=json.loads(l)
print("$MODE", "ms/step %.2f" % d["ms_per_step"], "loss", d["final_loss"], "ppx", d["perplexity"], "roofline", d["roofline"]["kernel"], "%.3f" % d["roofline"]["frac"])
print({k: round(v,2) for k,v in d["kernel_ms_per_step"].items()})
print("knm", d["roofline_knm"]["frac"], d["roofline_knm"]["in_step"]["frac"])
PY
done
exit $RC
