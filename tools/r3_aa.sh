python -m pytest tests/test_subsequence.py -m gpu -x -q 2>&1 | tail -5
