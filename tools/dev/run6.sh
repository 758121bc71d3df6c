timeout 1700 python -m pytest tests/test_igemm_variants.py -x -q -m gpu -p no:cacheprovider -k "patch or full_size" 2>&1 | tail -8
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'])
for c in d.get('configs',[]):
    print(json.dumps({k:c[k] for k in c if k in ('name','value','unit','ms_per_step','roofline','config')})[:600])
"
