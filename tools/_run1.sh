export TMPDIR=/tmp
rm -rf /tmp/kt_a; 
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_a -o runc -- python bench.py --child --steps 3 --warmup 1 > gpurun_out/r2_kt2.log 2>&1
python tools/kernel_trace_summary.py $(ls /tmp/kt_a/*kernel_trace.csv /tmp/kt_a/*/*kernel_trace.csv 2>/dev/null | tail -1) > gpurun_out/r2_kt2_summary.csv
for th in 1 2 3; do python bench.py --threads $th --no-pmc --no-cpu-baseline --steps 12 --warmup 3 --verify 200 > gpurun_out/r2_th$th.json 2> gpurun_out/r2_th$th.err; grep "PCIe" gpurun_out/r2_th$th.err; done
for hs in 131072 524288 1048576; do FMX_HOST_SUB_QUERIES=$hs python bench.py --threads 2 --no-pmc --no-cpu-baseline --steps 12 --warmup 3 --verify 200 > gpurun_out/r2_hs$hs.json 2> gpurun_out/r2_hs$hs.err; echo hs=$hs; grep "PCIe" gpurun_out/r2_hs$hs.err; done
head -25 gpurun_out/r2_kt2_summary.csv
