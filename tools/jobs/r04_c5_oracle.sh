timeout -k 10 250 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5_first_pivots_are_the_cpu" 2>&1 | tail -3
