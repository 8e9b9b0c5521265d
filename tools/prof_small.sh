#!/bin/bash
# rocprofv3 kernel stats of `bench.py --model small --lanes 1` -> gpurun_out/<tag>_bench_small_b64_lanes1_kernel_stats.csv
set -e -o pipefail
tag=${1:-r03}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_small -- python $root/bench.py --model small --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-secondary > $out/${tag}_prof_small.log 2>&1
cp $(find $out/${tag}_prof_small -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_small_b64_lanes1_kernel_stats.csv
rm -rf $out/${tag}_prof_small
