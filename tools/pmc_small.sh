#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes + the calibration pass of each) for `--model small`, merged into
# profiles-style JSON:  tools/pmc_small.sh <tag> <commit>  -> gpurun_out/<tag>_pmc_traffic_small.json
set -e -o pipefail
tag=${1:-r03}; commit=${2:-unknown}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmcs_$c -- python $root/bench.py --model small --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline --no-secondary > $out/${tag}_pmcs_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_cals_$c -- python $root/tools/pmc_calib.py > $out/${tag}_cals_$c.log 2>&1
  echo "pmc $c done"
done
cd $root
python tools/pmc_traffic.py $out/${tag}_pmcs_FETCH_SIZE $out/${tag}_pmcs_WRITE_SIZE $out/${tag}_pmc_traffic_small.json small-b64 $commit $out/${tag}_cals_FETCH_SIZE $out/${tag}_cals_WRITE_SIZE | head -12
rm -rf $out/${tag}_pmcs_FETCH_SIZE $out/${tag}_pmcs_WRITE_SIZE $out/${tag}_cals_FETCH_SIZE $out/${tag}_cals_WRITE_SIZE
