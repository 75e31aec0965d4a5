python -m pytest tests/test_gpu_dist.py tests/test_gpu_train.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -12
