export TMPDIR=/tmp
mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_device_entry.py -x -q > gpurun_out/r3m/tests.log 2>&1; tail -3 gpurun_out/r3m/tests.log
for t in 1 2; do
rm -rf /tmp/pl$t
FMX_CHILD_MODE=pipeline FMX_TURNS=$t rocprofv3 --kernel-trace --output-format csv -d /tmp/pl$t -o run -- python bench.py --child --steps 9 --warmup 3 > gpurun_out/r3m/pl$t.log 2>&1
python tools/host_busy.py analyse /tmp/pl$t 9 > gpurun_out/r3m/pl${t}_analysis.txt
grep window gpurun_out/r3m/pl${t}_analysis.txt
done
