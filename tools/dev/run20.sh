timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_line.json'))
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'])
for c in d.get('configs',[]):
    if 'error' in c: print(c); continue
    print(c['baseline_config'], '|', round(c['value']), c['unit'], '| ms/pass', round(c['ms_per_pass'],4), '| frac', round(c['mfma_frac_whole_set'],4))
print(json.dumps(d['cpu_baseline'])[:1800])
PY
timeout 900 python -m pytest tests/test_sharding.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
