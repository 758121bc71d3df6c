# same-box A/B of a bench.py pass between the built library and csi-nn2_amd/lib_base/libshl_prev.so: tools/dev/ab_lib.sh [reps] [bench.py arguments]
L=csi-nn2_amd/lib
cp $L/libshl_mi355x.so /tmp/cur.so
REPS=${1:-3}; shift
for rep in $(seq 1 $REPS); do
  for v in cur prev; do
    if [ $v = cur ]; then cp /tmp/cur.so $L/libshl_mi355x.so; else cp ${PREV:-csi-nn2_amd/lib_base/libshl_prev.so} $L/libshl_mi355x.so; fi
    echo -n "$v: "; python bench.py --no-cpu-baseline --no-configs "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f us' % (1000*d['ms_per_step']))"
  done
done
cp /tmp/cur.so $L/libshl_mi355x.so
