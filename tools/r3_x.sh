export TMPDIR=/tmp
mkdir -p gpurun_out/r3x
python -m pytest tests -m gpu -x -q > gpurun_out/r3x/tests.log 2>&1; tail -3 gpurun_out/r3x/tests.log
python bench.py --force-gather --no-pmc --no-cpu-baseline --no-text-leg 2> gpurun_out/r3x/fg.err > gpurun_out/r3x/fg.json; grep "value (inputs" gpurun_out/r3x/fg.err
