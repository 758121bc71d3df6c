export SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG=32
for lay in NCHW; do for l in 0 4 8 14; do python tools/pp_trace.py --patch --layer $l --layout $lay 2>&1 | grep "epilogue\|workgroups:"; done; done
