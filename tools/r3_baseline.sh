# round-3 baseline session: 125 k-query pass timeline + C3 bench (no pmc) + C2 bench
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt125 -o run -- python bench.py --child --queries 125000 --steps 3 --warmup 2 > gpurun_out/r3a/kt125.log 2>&1
python tools/timeline.py /tmp/kt125 1 0 > gpurun_out/r3a/timeline125.txt
python bench.py --queries 125000 --no-pmc --no-cpu-baseline > gpurun_out/r3a/b125.json 2> gpurun_out/r3a/b125.err
python bench.py --no-pmc --no-cpu-baseline > gpurun_out/r3a/c3.json 2> gpurun_out/r3a/c3.err
python bench.py --config C2 --no-pmc --no-cpu-baseline > gpurun_out/r3a/c2.json 2> gpurun_out/r3a/c2.err
tail -4 gpurun_out/r3a/*.err
