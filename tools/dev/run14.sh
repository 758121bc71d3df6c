timeout 1500 python -m pytest tests/test_igemm_variants.py -x -q -m gpu -p no:cacheprovider -k "patch" 2>&1 | grep -v "^$" | tail -60
