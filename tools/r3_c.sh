export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_multidev.py -x -q > gpurun_out/r3c/multidev.log 2>&1
tail -3 gpurun_out/r3c/multidev.log
for t in 0 1; do
rm -rf /tmp/hb$t
NQ=125000 FMX_TURNS=$t rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/hb$t -o run -- python tools/host_busy.py 3 30 > gpurun_out/r3c/hb$t.log 2>&1
grep round gpurun_out/r3c/hb$t.log
python tools/host_busy.py analyse /tmp/hb$t 60 4 > gpurun_out/r3c/hb${t}_analysis.txt
done
