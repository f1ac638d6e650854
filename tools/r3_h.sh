bash tools/ab.sh r3h "" main w7 w6
python -m pytest tests/test_facade.py tests/test_gpu_parity.py tests/test_gpu_layout.py -x -q > gpurun_out/r3h/tests.log 2>&1; tail -3 gpurun_out/r3h/tests.log
