export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
B="python bench.py --no-pmc --no-cpu-baseline --verify 500 --model-queries 0 --steps 16"
run() { tag=$1; shift; env "$@" $B 2> gpurun_out/r3d/$tag.err | cut -c1-300 > gpurun_out/r3d/$tag.json; echo "$tag: $(grep -h 'PCIe-inclusive' gpurun_out/r3d/$tag.err)"; }
run base_t3 FMX_X=0
run split1_t2 FMX_CU_SPLIT=1 FMX_TURNS=0 FMX_BENCH_THREADS=2
run split1_t4 FMX_CU_SPLIT=1 FMX_TURNS=0 FMX_BENCH_THREADS=4
run split2_t2 FMX_CU_SPLIT=2 FMX_TURNS=0 FMX_BENCH_THREADS=2
run split2_t4 FMX_CU_SPLIT=2 FMX_TURNS=0 FMX_BENCH_THREADS=4
run nosplit_t2 FMX_TURNS=0 FMX_BENCH_THREADS=2
run nosplit_t4 FMX_TURNS=0 FMX_BENCH_THREADS=4
python -m pytest tests/test_prior_matches.py tests/test_gpu_multidev.py -x -q > gpurun_out/r3d/tests.log 2>&1; tail -3 gpurun_out/r3d/tests.log
