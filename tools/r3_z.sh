export TMPDIR=/tmp
mkdir -p gpurun_out/r3z
python tools/time_build.py > gpurun_out/r3z/time_build.txt 2>&1; cat gpurun_out/r3z/time_build.txt
python -m pytest tests/test_gpu_scale.py tests/test_gpu_derive.py tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_abi.py tests/test_facade.py -m gpu -x -q 2>&1 | tail -3
