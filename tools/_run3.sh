export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --no-pmc --no-cpu-baseline --steps 6 --warmup 2 --verify 2000 > gpurun_out/r2_b3.json 2> gpurun_out/r2_b3.err; grep "kernel ms\|PCIe\|last step" gpurun_out/r2_b3.err
FMX_SERIAL=1 python bench.py --no-pmc --no-cpu-baseline --steps 4 --warmup 2 --verify 0 2>&1 >/dev/null | grep "kernel ms"
