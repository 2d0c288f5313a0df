#!/bin/bash
# Regenerates the files of profiles/rNN on the GPU box (run through gpurun from the repo root): tools/profile_round.sh <out dir under gpurun_out>
set -e
R=$PWD; O=$R/gpurun_out/${1:-prof_round}; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $O/bench_default_run.log 2>$O/bench_default_run.err
echo "default run done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-baseline-n 0 > $O/bench_under_trace.log 2>&1
echo "trace done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py --steps 1 --warmup 0 --knm-iters 3 --cpu-baseline-n 0 > $O/pmc_w.log 2>&1
echo "write pmc done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py --steps 1 --warmup 0 --knm-iters 3 --cpu-baseline-n 0 > $O/pmc_f.log 2>&1
echo "fetch pmc done"
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
grep '^{' $O/bench_default_run.log | tail -1 | cut -c1-300
