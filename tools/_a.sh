timeout -k 10 120 tools/ubench/attn_loop | grep -E "^full kernel  |V\^T|skeleton"
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pa -- $root/tools/ubench/attn_loop > $root/gpurun_out/pa.log 2>&1
python $root/tools/pmc_kernel.py $root/gpurun_out/pa attn_kernel | grep -A2 "ILi2ELi0E\|ILi2ELi32E\|<2, 0>\|<2, 32>"
rm -rf $root/gpurun_out/pa
