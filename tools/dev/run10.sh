timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
