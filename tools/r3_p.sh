export TMPDIR=/tmp
mkdir -p gpurun_out/r3p
rm -rf /tmp/bt
FMX_TURNS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -o run -- python bench.py --no-pmc --no-cpu-baseline --verify 0 --model-queries 0 --steps 12 > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err
grep "value (inputs" gpurun_out/r3p/bench.err
python tools/passes.py /tmp/bt 9 12 > gpurun_out/r3p/passes.txt
cat gpurun_out/r3p/passes.txt
