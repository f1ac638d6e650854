# design-time: kernel + memory-copy trace of the value leg (text batches, four in flight); timeline of the last passes
export TMPDIR=/tmp
rm -rf /tmp/tt
FMX_CHILD_MODE=text rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tt -o run -- python bench.py --child --steps 8 --warmup 4 > gpurun_out/tt.log 2>&1
python tools/dbg/timeline.py /tmp/tt 6 40 > gpurun_out/tt_timeline.txt
wc -l gpurun_out/tt_timeline.txt
