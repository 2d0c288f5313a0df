#!/bin/bash
# PMC passes of the bench command (one counter group per pass, --kernel-trace only: the pool refuses --pmc with other trace domains).
# usage (through gpurun, repo root): tools/pmc_round.sh <out dir under gpurun_out> [extra bench args]
R=$PWD; O=$R/gpurun_out/${1:-pmc}; shift; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="--steps 1 --warmup 1 --knm-iters 2 --kernel-pass-steps 1 --cpu-baseline-n 0 $@"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench.py $B > $O/$name.log 2>&1 || { echo "$name failed"; tail -3 $O/$name.log; exit 1; }; echo "$name done"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE
run f FETCH_SIZE
run w WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
for d in sq lds f w tcc; do python3 $R/tools/pmc_summary.py $O/$d > $O/$d.csv; done
head -40 $O/sq.csv | cut -c1-240
