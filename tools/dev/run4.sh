export SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32
for lay in NHWC NCHW; do for l in 4 14; do python tools/pp_trace.py --patch --layer $l --layout $lay 2>&1 | tail -17; done; done
