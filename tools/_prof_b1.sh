set -e
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp; cd /tmp
for m in small tiny; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/pb1_$m -- python $root/bench.py --model $m --batch 1 --lanes 1 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > $out/pb1_$m.log 2>&1
cp $(find $out/pb1_$m -name "*kernel_stats.csv" | head -1) $out/pb1_${m}_kernel_stats.csv
rm -rf $out/pb1_$m
done
