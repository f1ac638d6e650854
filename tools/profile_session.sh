# One profiling session on the GPU box; raw output under gpurun_out/prof_<cfg>/, summaries by tools/profile_refresh.py.
#   bash tools/profile_session.sh C3 [full]     (full: also the SQ counter pass)
export TMPDIR=/tmp
cfg=${1:-C3}
mode=${2:-}
out=gpurun_out/prof_$cfg
rm -rf $out; mkdir -p $out
timeout 600 python bench.py --config $cfg > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
child="python bench.py --config $cfg --child --steps 4 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o run -- $child > $out/kt.log 2>&1
# the timed region itself (calls overlapped on the GPU): busy time of the GPU, kernel durations while they share it
FMX_CHILD_MODE=pipeline timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_pipeline -o run -- python bench.py --config $cfg --child --steps 9 --warmup 3 > $out/kt_pipeline.log 2>&1
python tools/host_busy.py analyse $out/kt_pipeline 9 > $out/pipeline_busy.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o run -- $child > $out/pmc_$c.log 2>&1
done
if [ "$mode" = "full" ]; then
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_SQ -o run -- $child > $out/pmc_SQ.log 2>&1
fi
# keep only the CSVs the summariser reads (the merge back is capped at 64 MiB)
find $out -type f ! -name '*kernel_trace.csv' ! -name '*counter_collection.csv' ! -name 'bench.*' ! -name '*.log' ! -name 'pipeline_busy.txt' -delete
rm -f $out/kt_pipeline/*kernel_trace.csv $out/kt_pipeline/*/*kernel_trace.csv
du -sh $out
