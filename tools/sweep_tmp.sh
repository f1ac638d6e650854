run() { tag=$1; shift; env "$@" python bench.py --no-pmc --no-cpu-baseline --model-queries 0 --verify 0 --steps 8 > gpurun_out/s4_sw_$tag.json 2> gpurun_out/s4_sw_$tag.err; echo "== $tag $@"; grep "PCIe-inclusive\|kernel ms\|last step" gpurun_out/s4_sw_$tag.err; }
run base X=1
run lm1024 FMX_LIGHT_MAX=1024
run hg4096 FMX_HEAVY_GRID=4096
run hg2048 FMX_HEAVY_GRID=2048
run lm1024hg4096 FMX_LIGHT_MAX=1024 FMX_HEAVY_GRID=4096
run lm1024hg3072 FMX_LIGHT_MAX=1024 FMX_HEAVY_GRID=3072
