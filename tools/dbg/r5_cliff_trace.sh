# design-time: kernel trace of the value leg with SIX text calls in flight (the "cliff"): are the kernels slow or are there gaps?
export TMPDIR=/tmp
rm -rf /tmp/tc
FMX_CHILD_MODE=text FMX_SERIAL=1 FMX_TURN_MIN_HOST=1000000000 FMX_MAX_HOST_ACTIVE=6 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tc -o run -- python bench.py --child --steps 6 --warmup 2 --text-calls 6 --host-threads 6 --threads 8 > gpurun_out/tc.log 2>&1
python tools/dbg/timeline.py /tmp/tc 8 40 > gpurun_out/tc_timeline.txt
wc -l gpurun_out/tc_timeline.txt
