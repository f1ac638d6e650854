export TMPDIR=/tmp
mkdir -p gpurun_out/r3o
rm -rf /tmp/bt
FMX_TURNS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -o run -- python bench.py --no-pmc --no-cpu-baseline --verify 0 --model-queries 0 --steps 12 > gpurun_out/r3o/bench.json 2> gpurun_out/r3o/bench.err
grep "value (inputs" gpurun_out/r3o/bench.err
python tools/gaps.py /tmp/bt 300 > gpurun_out/r3o/gaps.txt
head -60 gpurun_out/r3o/gaps.txt
