#!/bin/bash
# rocprofv3 kernel stats of tools/ar_bench.py -> gpurun_out/<tag>_ar_kernel_stats.csv (+ the bench JSON)
set -e -o pipefail
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python tools/ar_bench.py > $out/${tag}_ar_decode.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_ar -- python $root/tools/ar_bench.py > /dev/null 2>&1
cp $(find $out/${tag}_prof_ar -name "*kernel_stats.csv" | head -1) $out/${tag}_ar_kernel_stats.csv
rm -rf $out/${tag}_prof_ar
cat $out/${tag}_ar_decode.json
head -12 $out/${tag}_ar_kernel_stats.csv | cut -c1-60,100-400
