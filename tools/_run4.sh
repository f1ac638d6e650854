export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for th in 1 2; do python bench.py --threads $th --no-pmc --no-cpu-baseline --steps 12 --warmup 3 --verify 500 > gpurun_out/r2_b4_$th.json 2> gpurun_out/r2_b4_$th.err; grep "PCIe" gpurun_out/r2_b4_$th.err; done
for hs in 131072 524288; do FMX_HOST_SUB_QUERIES=$hs python bench.py --threads 2 --no-pmc --no-cpu-baseline --steps 12 --warmup 3 --verify 0 2>&1 >/dev/null | grep PCIe; done
