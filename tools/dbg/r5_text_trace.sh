# design-time: kernel + memory-copy trace of the value leg (text batches, five in flight); timeline of the last passes and of the start
export TMPDIR=/tmp
rm -rf /tmp/tt
FMX_MAX_HOST_ACTIVE=5 FMX_CHILD_MODE=text rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tt -o run -- python bench.py --child --steps 10 --warmup 4 > gpurun_out/tt.log 2>&1
python tools/dbg/timeline.py /tmp/tt 6 40 > gpurun_out/tt_timeline.txt
python tools/dbg/fill_timeline.py /tmp/tt 100 40 > gpurun_out/tt_fill.txt
wc -l gpurun_out/tt_timeline.txt gpurun_out/tt_fill.txt
