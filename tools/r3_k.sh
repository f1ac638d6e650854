export TMPDIR=/tmp
mkdir -p gpurun_out/r3k
python -m pytest tests -m gpu -x -q > gpurun_out/r3k/tests.log 2>&1; tail -3 gpurun_out/r3k/tests.log
bash tools/ab.sh r3k "" main
FMX_TURNS=2 bash tools/ab.sh r3k_t2 "" main
