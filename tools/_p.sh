root=$(pwd); cd /tmp; export TMPDIR=/tmp
for cfg in "small_b64 --model small" "small_b1 --model small --batch 1" "base_b32 --model base --batch 32"; do
set -- $cfg; name=$1; shift
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $root/gpurun_out/pk -- python $root/bench.py "$@" --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline --no-secondary > $root/gpurun_out/pk.log 2>&1
python $root/tools/pmc_kernel.py $root/gpurun_out/pk > $root/gpurun_out/lds_$name.txt
rm -rf $root/gpurun_out/pk
echo $name done
done
