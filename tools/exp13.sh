run() { env $1 python bench.py --no-pmc --no-cpu-baseline --no-text-leg --model-queries 0 $2 2>gpurun_out/last.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'], 'pcie', d['pcie_inclusive']['ms_per_step'], d['pcie_inclusive']['value'], 'staged', d['staged']['ms_per_step'], d['parity']['mismatches'])"; grep "host-path calls" gpurun_out/last.err | sed 's/.*retries/retries/'; }
for i in 1 2 3; do run A=0 ""; done
for i in 1 2; do run A=0 "--config C2"; done
for i in 1 2; do run A=0 "--queries 125000"; done
run A=0 "--config C4"
run A=0 "--config C5"
run A=0 "--threads 8"
run A=0 "--config C2 --threads 8"
