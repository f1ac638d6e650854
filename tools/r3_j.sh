export TMPDIR=/tmp
mkdir -p gpurun_out/r3j
B="python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 24"
run() { tag=$1; shift; env "$@" $B $ARGS 2> gpurun_out/r3j/$tag.err > gpurun_out/r3j/$tag.json; echo "$tag: $(grep -h 'PCIe-inclusive' gpurun_out/r3j/$tag.err | sed 's/\[bench\] //') $(python -c "import json;print(json.load(open('gpurun_out/r3j/$tag.json'))['parity']['mismatches'])")"; }
ARGS=""
run c3_t1 FMX_TURNS=1
run c3_t2 FMX_TURNS=2
run c3_t2_th2 FMX_TURNS=2 FMX_BENCH_THREADS=2
run c3_t2_th4 FMX_TURNS=2 FMX_BENCH_THREADS=4
ARGS="--config C2"
run c2_t1 FMX_TURNS=1
run c2_t2 FMX_TURNS=2
ARGS="--queries 125000"
run q125_t1 FMX_TURNS=1
run q125_t2 FMX_TURNS=2
ARGS="--config C5"
run c5_t1 FMX_TURNS=1
run c5_t2 FMX_TURNS=2
