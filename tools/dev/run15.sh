timeout 900 python -m pytest tests/test_channel_ops.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
