set -u
mkdir -p gpurun_out/r02_e
O=gpurun_out/r02_e
timeout 900 python -m pytest tests/test_igemm_variants.py -m gpu -q -p no:cacheprovider -k "IGEMM=pp" > $O/pp_tests.log 2>&1; tail -5 $O/pp_tests.log
timeout 600 python -m pytest tests/test_channel_ops.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x > $O/chan_parity.log 2>&1; tail -5 $O/chan_parity.log
for f in 256x128 256x128x2 256x256; do
  SHL_MI355X_IGEMM=pp SHL_MI355X_PP=$f timeout 300 python tools/kbench.py --set resnet --batch 128 --layers 3,4,7,8,13,14 > $O/kb_pp_$f.log 2>&1; echo "== forced $f"; tail -7 $O/kb_pp_$f.log
done
for f in 256x128; do
for d in 2 46; do
  SHL_MI355X_DEBUG=$d SHL_MI355X_IGEMM=pp SHL_MI355X_PP=$f timeout 300 python tools/kbench.py --set resnet --batch 128 --layers 4,8 > $O/kb_${f}_dbg$d.log 2>&1; echo "== $f debug $d"; tail -3 $O/kb_${f}_dbg$d.log | head -2
done; done
for f in 256x128 256x128x2 256x256; do
  for l in 4 8; do
  SHL_MI355X_DEBUG=128 SHL_MI355X_IGEMM=pp SHL_MI355X_PP=$f timeout 120 python tools/pp_trace.py --layer $l > $O/trace_${f}_l$l.log 2>&1; echo "== trace $f layer $l"; cat $O/trace_${f}_l$l.log
  done
done
