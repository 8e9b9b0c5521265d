set -e
python -m pytest tests/test_gpu_ops.py tests/test_gpu_dit.py tests/test_gpu_fused.py tests/test_gpu_edges.py -q -x > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/sw.json 2>gpurun_out/sw_err.log; python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_class']['attention'])"
for m in tiny small base; do python bench.py --model $m --batch 1 --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/sw.json 2>>gpurun_out/sw_err.log; python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('$m b1', d['ms_per_step'])"; done
