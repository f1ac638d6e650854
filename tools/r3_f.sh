export TMPDIR=/tmp
mkdir -p gpurun_out/r3f
python -m pytest tests/test_facade.py -x -q > gpurun_out/r3f/facade.log 2>&1; tail -3 gpurun_out/r3f/facade.log
for k in 0 1 2 3 4 5; do
FMX_DBG_STOP=$k python bench.py --no-pmc --no-cpu-baseline --verify 0 --model-queries 0 --steps 4 --warmup 1 > /dev/null 2> gpurun_out/r3f/stop$k.err
echo "stop $k: $(grep -h 'kernel ms' gpurun_out/r3f/stop$k.err | sed 's/.*expand_l.: \([0-9.]*\).*/expand_l \1/')"
done
