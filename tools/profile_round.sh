#!/bin/bash
# Collects the judged evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>            e.g. r01_d
# 1. bench.py (default command) -> gpurun_out/<tag>_bench_tiny_b64.json
# 2. rocprofv3 --kernel-trace --stats of the same command        -> <tag>_bench_tiny_b64_kernel_stats.csv
#    and of the single-lane command (no cross-stream overlap in the per-kernel durations) -> ..._lanes1_kernel_stats.csv
# 3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) -> <tag>_pmc_traffic.json
# Copy the files you want judged from gpurun_out/ into profiles/.
set -e -o pipefail
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench_tiny_b64.json 2> $out/${tag}_bench.err
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof.log 2>&1
cp $(find $out/${tag}_prof -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_tiny_b64_kernel_stats.csv
echo "rocprof (default lanes) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof1 -- python $root/bench.py --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline > $out/${tag}_prof1.log 2>&1
cp $(find $out/${tag}_prof1 -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_tiny_b64_lanes1_kernel_stats.csv
echo "rocprof (1 lane) done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_fetch -- python $root/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline > $out/${tag}_pmcf.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_write -- python $root/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline > $out/${tag}_pmcw.log 2>&1
echo "pmc write done"
cd $root
python tools/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_traffic.json
# keep the merged-back payload small: drop the raw traces
rm -rf $out/${tag}_prof $out/${tag}_prof1 $out/${tag}_pmc_fetch $out/${tag}_pmc_write
head -c 600 $out/${tag}_bench_tiny_b64.json; echo
head -8 $out/${tag}_bench_tiny_b64_lanes1_kernel_stats.csv
