export TMPDIR=/tmp
rm -rf /tmp/kt_a
FMX_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_a -o runc -- python bench.py --child --steps 3 --warmup 1 > gpurun_out/r2_kt3.log 2>&1
python tools/kernel_trace_summary.py $(ls /tmp/kt_a/*kernel_trace.csv /tmp/kt_a/*/*kernel_trace.csv 2>/dev/null | tail -1) > gpurun_out/r2_kt3_summary.csv
grep -v rocprim gpurun_out/r2_kt3_summary.csv | head -12
timeout 900 python -m pytest tests/test_subsequence.py -x -q -m gpu 2>&1 | tail -15
