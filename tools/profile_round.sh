#!/bin/bash
# Collects the judged evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> <commit>            e.g. tools/profile_round.sh r02_a $(git rev-parse --short HEAD)
# 1. bench.py (default command, incl. secondary lines and cpu_baseline) -> gpurun_out/<tag>_bench_tiny_b64.json
# 2. rocprofv3 --kernel-trace --stats of the same workload, two lanes and one lane (serial kernels: the averages
#    roofline.avg_launch_ms must agree with), and of --model small                        -> *_kernel_stats.csv
# 3. per-shape GEMM tables (SVC_PROF_DUMP)                                                 -> *_gemm_shapes_*.txt
# 4. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) + the 1 GiB calibration pass of each counter
#                                                                  -> <tag>_pmc_traffic.json + <tag>_pmc_traffic_per_kernel.csv
# 5. AR decode bench + its kernel stats, the a16 seam kernel's bandwidth table + its kernel stats, B = 1 latency lines, base model line.
# 6. one SQ-counter pass (MFMA busy, LDS, waits) for the fused / kconv / attention kernels   -> <tag>_pmc_sq_counters.txt
# Copy the files you want judged from gpurun_out/ into profiles/.
set -e -o pipefail
tag=${1:-r02}
commit=${2:-unknown}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench_tiny_b64.json 2> $out/${tag}_bench.err
echo "bench done"
python bench.py --model small --no-cpu-baseline --no-secondary > $out/${tag}_bench_small_b64.json 2>> $out/${tag}_bench.err
python bench.py --model base --batch 32 --no-cpu-baseline --no-secondary > $out/${tag}_bench_base_b32.json 2>> $out/${tag}_bench.err
python bench.py --model v2 --no-cpu-baseline --no-secondary > $out/${tag}_bench_v2_b64.json 2>> $out/${tag}_bench.err
python bench.py --model v2 --batch 1 --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/${tag}_bench_v2_b1.json 2>> $out/${tag}_bench.err
for m in tiny small base; do python bench.py --model $m --batch 1 --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/${tag}_bench_${m}_b1.json 2>> $out/${tag}_bench.err; done
echo "model lines done"
python tools/ar_bench.py > $out/${tag}_ar_decode.json 2>> $out/${tag}_bench.err
python tools/aa_act_bench.py > $out/${tag}_aa_act.json 2>> $out/${tag}_bench.err
shapes() {  # name, then bench.py arguments
  local name=$1; shift
  rm -f $out/shapes.csv
  SVC_PROF_DUMP=$out/shapes.csv python bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python tools/shape_report.py $out/shapes.csv > $out/${tag}_gemm_shapes_$name.txt
}
shapes tiny_b64 --model tiny
shapes small_b64 --model small
shapes base_b32 --model base --batch 32
shapes tiny_b1 --model tiny --batch 1 --lanes 1
shapes small_b1 --model small --batch 1 --lanes 1
rm -f $out/shapes.csv
cd /tmp
prof() {  # name, then bench.py arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_$name -- python $root/bench.py "$@" > $out/${tag}_prof_$name.log 2>&1
  cp $(find $out/${tag}_prof_$name -name "*kernel_stats.csv" | head -1) $out/${tag}_${name}_kernel_stats.csv
  rm -rf $out/${tag}_prof_$name
}
prof bench_tiny_b64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
prof bench_tiny_b64_lanes1 --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-secondary
prof bench_small_b64_lanes1 --model small --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-secondary
prof bench_small_b1 --model small --batch 1 --lanes 1 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary
echo "rocprof kernel stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_ar -- python $root/tools/ar_bench.py > /dev/null 2>&1
cp $(find $out/${tag}_prof_ar -name "*kernel_stats.csv" | head -1) $out/${tag}_ar_kernel_stats.csv
rm -rf $out/${tag}_prof_ar
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_aa -- python $root/tools/aa_act_bench.py > /dev/null 2>&1
cp $(find $out/${tag}_prof_aa -name "*kernel_stats.csv" | head -1) $out/${tag}_aa_act_kernel_stats.csv
rm -rf $out/${tag}_prof_aa
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_$c -- python $root/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline --no-secondary > $out/${tag}_pmc_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_cal_$c -- python $root/tools/pmc_calib.py > $out/${tag}_cal_$c.log 2>&1
  echo "pmc $c done"
done
# SQ counters of the three dominant kernels (one more --pmc pass, --kernel-trace only): MFMA busy, LDS activity / conflicts, waits
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/${tag}_pmc_sq -- python $root/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline --no-secondary > $out/${tag}_pmc_sq.log 2>&1
for k in dit_panel_kernel kconv_kernel attn_kernel; do python $root/tools/pmc_kernel.py $out/${tag}_pmc_sq $k; done > $out/${tag}_pmc_sq_counters.txt
rm -rf $out/${tag}_pmc_sq
echo "pmc sq done"
cd $root
python tools/pmc_traffic.py $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_traffic.json tiny-b64 $commit $out/${tag}_cal_FETCH_SIZE $out/${tag}_cal_WRITE_SIZE
# keep the merged-back payload small: drop the raw per-dispatch traces (the per-kernel aggregate CSV stays)
rm -rf $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_cal_FETCH_SIZE $out/${tag}_cal_WRITE_SIZE
head -c 700 $out/${tag}_bench_tiny_b64.json; echo
head -8 $out/${tag}_bench_tiny_b64_lanes1_kernel_stats.csv | cut -c1-160
