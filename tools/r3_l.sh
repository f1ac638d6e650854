export TMPDIR=/tmp
mkdir -p gpurun_out/r3l
python -m pytest tests/test_gpu_device_entry.py tests/test_gpu_bench_multi.py -x -q > gpurun_out/r3l/tests.log 2>&1; tail -3 gpurun_out/r3l/tests.log
for t in 1 2; do FMX_TURNS=$t bash tools/ab.sh r3l_t$t "" main; done
FMX_TURNS=2 bash tools/ab.sh r3l_c2 "--config C2" main
FMX_TURNS=2 bash tools/ab.sh r3l_q125 "--queries 125000" main
