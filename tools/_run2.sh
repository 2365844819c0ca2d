timeout -k 10 600 python -m pytest tests/test_gt_parity_gpu.py -x -q -m gpu -k "runs_of_lines" 2>&1 | tail -4
