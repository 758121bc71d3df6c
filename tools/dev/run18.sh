timeout 900 python bench.py --detail 2> gpurun_out/bench_detail.txt | tail -1 > gpurun_out/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_line.json'))
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'])
for c in d.get('configs',[]):
    if 'error' in c: print(c); continue
    print(c['baseline_config'], '|', round(c['value']), c['unit'], '| ms/pass', round(c['ms_per_pass'],4), '| frac', round(c['mfma_frac_whole_set'],4), '| kernel_only', c['kernel_only'], c.get('relayout_layers'))
print(json.dumps(d['cpu_baseline'])[:1500])
PY
timeout 600 python bench.py --workload resnet50_3x3 --total-batch 128 --steps 5 --warmup 2 --windows 3 --no-cpu-baseline --no-configs | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['parallelism'])"
timeout 900 python -m pytest tests/test_sharding.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
