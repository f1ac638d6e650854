export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
python bench.py --no-pmc --no-cpu-baseline --verify 0 --model-queries 0 --steps 2 --warmup 1 2> gpurun_out/r3e/hist.err > gpurun_out/r3e/hist.json
grep "light hist" gpurun_out/r3e/hist.err | tail -2
python -m pytest tests/test_facade.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3
