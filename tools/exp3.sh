run() { python bench.py --no-pmc --no-cpu-baseline --no-text-leg --model-queries 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'], 'pcie', d['pcie_inclusive']['ms_per_step'], 'staged', d['staged']['ms_per_step'], d['parity']['mismatches'])"; }
run --queries 125000
run --config C2
run
run --queries 125000
run --config C2
