timeout 1200 python -m pytest tests/test_channel_ops.py tests/test_sharding.py tests/test_gpu_parity.py tests/test_tail.py tests/test_whole_network.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -12
