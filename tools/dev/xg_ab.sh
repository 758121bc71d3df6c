# A/B of the XCD-aware workgroup mapping of pwdw_fused: pass time + HBM traffic (PMC) per setting
export TMPDIR=/tmp
REPO=$PWD
for xg in 0 auto 1; do
  if [ $xg = auto ]; then unset SHL_MI355X_PWDW_XG; else export SHL_MI355X_PWDW_XG=$xg; fi
  echo "== SHL_MI355X_PWDW_XG=$xg"
  python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('img/s', d['value'], 'ms', d['ms_per_step'])"
  OUT=$REPO/gpurun_out/xg_$xg; mkdir -p $OUT; cd /tmp
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o t -- python "$REPO/bench.py" --steps-only --steps 20 --warmup 2 --windows 1 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o t -- python "$REPO/bench.py" --steps-only --steps 20 --warmup 2 --windows 1 > /dev/null 2>&1
  cd $REPO
  python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict): print('  ',k, v['dispatches'], round(v['hbm_bytes_per_launch']))"
  rm -rf $OUT
done
