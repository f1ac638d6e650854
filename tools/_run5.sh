export TMPDIR=/tmp
for sq in 131072 262144; do echo sub=$sq; FMX_SUB_QUERIES=$sq python bench.py --threads 2 --no-pmc --no-cpu-baseline --steps 6 --warmup 2 --verify 0 2>&1 >/dev/null | grep "kernel ms\|PCIe"; done
