export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
python -m pytest tests/test_gpu_device_entry.py tests/test_gpu_parity.py -x -q > gpurun_out/r3q/tests.log 2>&1; tail -3 gpurun_out/r3q/tests.log
for t in 1 2; do FMX_TURNS=$t bash tools/ab.sh r3q_t$t "" main; done
FMX_TURNS=2 FMX_BENCH_THREADS=2 bash tools/ab.sh r3q_t2th2 "" main
FMX_TURNS=2 bash tools/ab.sh r3q_c2 "--config C2" main
FMX_TURNS=2 bash tools/ab.sh r3q_q125 "--queries 125000" main
FMX_TURNS=1 bash tools/ab.sh r3q_q125t1 "--queries 125000" main
