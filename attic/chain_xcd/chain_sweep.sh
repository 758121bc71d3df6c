for from in 99 1 3 5 7 9; do for wgs in 32 16; do
  echo "== CHAIN_FROM=$from WGS=$wgs"
  SHL_MI355X_CHAIN_FROM=$from SHL_MI355X_CHAIN_WGS=$wgs timeout 200 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"
done; done
