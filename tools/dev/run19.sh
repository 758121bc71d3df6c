timeout 600 python bench.py --workload resnet50_3x3 --total-batch 128 --steps 5 --warmup 2 --windows 3 --no-cpu-baseline --no-configs 2>&1 | tail -15 | cut -c1-1500
