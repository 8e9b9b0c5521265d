set -e
python -m pytest tests/test_gpu_vocoder.py -q -x > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/sw.json 2>gpurun_out/sw_err.log; python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_class']['kgemm_f16'])"
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $root/gpurun_out/pk -- python $root/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline --no-secondary > $root/gpurun_out/pk.log 2>&1
python $root/tools/pmc_kernel.py $root/gpurun_out/pk kconv_kernel | head -6
rm -rf $root/gpurun_out/pk
