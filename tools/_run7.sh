export TMPDIR=/tmp
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for th in 2 3; do echo th=$th; FMX_HOST_SUB_QUERIES=1048576 python bench.py --threads $th --no-pmc --no-cpu-baseline --steps 12 --warmup 3 --verify 0 2>&1 >/dev/null | grep "PCIe\|index built\|Error\|error"; done
