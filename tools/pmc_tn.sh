#!/bin/bash
# FETCH_SIZE + duration of the TN launches for a given GDRF_TN_NSPLIT (exported by the caller)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for ns in "$@"; do
  export GDRF_TN_NSPLIT=$ns
  rm -rf $R/gpurun_out/pmc_tn_$ns
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tn_$ns -- python3 $R/bench.py --steps 1 --warmup 0 --knm-iters 1 --cpu-baseline-n 0 > $R/gpurun_out/pmc_tn_$ns.log 2>&1 || exit 1
  f=$(ls $R/gpurun_out/pmc_tn_$ns/*/*counter_collection.csv | head -1)
  echo "ns=$ns"; grep "tn_bf16x6" $f | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print('  grid', int(r[6])//256, round(float(r[16])*1024/1e9*2,1), 'GB(x2)', (int(r[18])-int(r[17]))/1e6,'ms')"
done
