export SHL_MI355X_IGEMM=patch SHL_MI355X_DEBUG_GEOM=1 SHL_MI355X_PATCH_PAIR=1
timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider -s -k "test_forced_variant_int8" 2>&1 | grep "patch geom" | grep -v "pair 0" | sort | uniq -c | head -20
