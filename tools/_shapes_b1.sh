rm -f gpurun_out/shapes.csv
SVC_PROF_DUMP=gpurun_out/shapes.csv python bench.py --model small --batch 1 --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/shape_report.py gpurun_out/shapes.csv > gpurun_out/shapes_small_b1.txt
rm -f gpurun_out/shapes.csv
SVC_PROF_DUMP=gpurun_out/shapes.csv python bench.py --model tiny --batch 1 --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/shape_report.py gpurun_out/shapes.csv > gpurun_out/shapes_tiny_b1.txt
