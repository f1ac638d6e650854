# design-time: what the GPU does while a pipeline of five text calls fills (the first call completes after ~20 ms, not ~10)
export TMPDIR=/tmp
rm -rf /tmp/tf
REPS=2 SLEEP_BETWEEN=0.3 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tf -o run -- python tools/dbg/text_host_time.py 5 > gpurun_out/tf.log 2>&1
python tools/dbg/fill_timeline.py /tmp/tf 150 26 > gpurun_out/tf_fill.txt
grep "queries,\|done at" gpurun_out/tf.log | cut -c1-200
wc -l gpurun_out/tf_fill.txt
