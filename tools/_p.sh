root=$(pwd); cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES"; do
i=$((i+1))
rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/pk -- python $root/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-roofline --no-secondary > $root/gpurun_out/pk.log 2>&1
for k in dit_panel_kernel kconv_kernel attn_kernel; do python $root/tools/pmc_kernel.py $root/gpurun_out/pk $k | head -10; done > $root/gpurun_out/sq_$i.txt
rm -rf $root/gpurun_out/pk
echo set $i done
done
cd $root
(python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/clk_bench.json 2>/dev/null &) 
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4; sleep 1; done > gpurun_out/clk.txt
wait
sleep 5
