for bm in 0 64 128; do
for m in tiny small; do
    SVC_KCONV_BM=$bm python bench.py --model $m --batch 1 --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/sw_${m}.json 2>gpurun_out/sw_err.log && python -c "
import json,sys; d=json.load(open('gpurun_out/sw_${m}.json')); print('$bm $m',d['ms_per_step'])"
done; done
