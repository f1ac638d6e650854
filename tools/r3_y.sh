bash tools/ab.sh r3y "--no-text-leg" main hw5 hw4
python -m pytest tests/test_gpu_scale.py -x -q 2>&1 | tail -2
