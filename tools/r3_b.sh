export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_multidev.py -x -q > gpurun_out/r3b/multidev.log 2>&1
tail -5 gpurun_out/r3b/multidev.log
for t in 0 1; do
FMX_TURNS=$t python bench.py --queries 125000 --no-pmc --no-cpu-baseline --verify 0 --model-queries 0 2> gpurun_out/r3b/b125_t$t.err | cut -c1-200 > gpurun_out/r3b/b125_t$t.json
FMX_TURNS=$t python bench.py --config C2 --no-pmc --no-cpu-baseline --verify 0 --model-queries 0 2> gpurun_out/r3b/c2_t$t.err | cut -c1-200 > gpurun_out/r3b/c2_t$t.json
done
grep -h "PCIe-inclusive" gpurun_out/r3b/*.err
