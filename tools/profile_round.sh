#!/bin/bash
# Regenerates the files of profiles/rNN on the GPU box (run through gpurun from the repo root): tools/profile_round.sh <out dir under gpurun_out>
set -e
R=$PWD; O=$R/gpurun_out/${1:-prof_round}; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 560 python3 $R/bench.py > $O/bench_default_run.log 2>$O/bench_default_run.err
echo "default run done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-baseline-n 0 > $O/bench_under_trace.log 2>&1
echo "trace done"
P="--steps 1 --warmup 1 --knm-iters 3 --kernel-pass-steps 1 --cpu-baseline-n 0"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py $P > $O/pmc_w.log 2>&1
echo "write pmc done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py $P > $O/pmc_f.log 2>&1
echo "fetch pmc done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py $P > $O/pmc_sq.log 2>&1
echo "sq pmc done"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/clk -- python3 $R/bench.py $P > $O/pmc_clk.log 2>&1
echo "clock pmc done"
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
for d in w f sq clk; do python3 $R/tools/pmc_summary.py $O/$d > $O/bench_pmc_$d.csv; done
python3 $R/tools/pmc_to_json.py $O $O/bench_default_run.log > $O/pmc_hbm.json
grep '^{' $O/bench_default_run.log | tail -1 | cut -c1-400
