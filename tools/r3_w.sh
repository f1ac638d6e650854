bash tools/ab.sh r3w "--no-text-leg" main lw6
python -m pytest tests/test_gpu_parity.py tests/test_gpu_layout.py tests/test_gpu_small.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
