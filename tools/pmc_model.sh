#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes + the calibration pass of each) for one `--model`, as a
# profiles-style JSON:  tools/pmc_model.sh <tag> <commit> <model> <workload key> [bench.py args]
#   e.g. tools/pmc_model.sh r03_k abc1234 base base-b32 --batch 32   -> gpurun_out/<tag>_pmc_traffic_<model>.json
set -e -o pipefail
tag=${1:-r03}; commit=${2:-unknown}; model=${3:-small}; key=${4:-small-b64}; shift 4 || true
root=$(pwd); out=$root/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmcm_$c -- python $root/bench.py --model $model "$@" --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline --no-secondary > $out/${tag}_pmcm_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_calm_$c -- python $root/tools/pmc_calib.py > $out/${tag}_calm_$c.log 2>&1
  echo "pmc $model $c done"
done
cd $root
python tools/pmc_traffic.py $out/${tag}_pmcm_FETCH_SIZE $out/${tag}_pmcm_WRITE_SIZE $out/${tag}_pmc_traffic_${model}.json $key $commit $out/${tag}_calm_FETCH_SIZE $out/${tag}_calm_WRITE_SIZE | head -12
rm -rf $out/${tag}_pmcm_FETCH_SIZE $out/${tag}_pmcm_WRITE_SIZE $out/${tag}_calm_FETCH_SIZE $out/${tag}_calm_WRITE_SIZE
