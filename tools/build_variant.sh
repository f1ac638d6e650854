# usage: bash tools/build_variant.sh <tag> [-DFOO=1 ...]  -> build/libfmx_<tag>.so (design-time A/B builds; select with FMX_LIB_PATH)
set -e
tag=$1; shift
cd "$(dirname "$0")/../fuzzy-match_amd/csrc"
mkdir -p ../../build
F="-O3 -std=c++17 -fPIC -DNDEBUG -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -fno-gpu-flush-denormals-to-zero "$@" -c fmx_gpu.hip -o ../../build/fmx_gpu_$tag.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build/libfmx_$tag.so fmx_host.o fmx_capi.o ../../build/fmx_gpu_$tag.o fmx_sort.o fmx_text.o -pthread -ldl -Wl,-rpath,/opt/rocm/lib
python ../../tools/kernel_resources.py ../../build/fmx_gpu_$tag.o > ../../build/resources_$tag.csv
rm -f ../../build/fmx_gpu_$tag.o
echo "built build/libfmx_$tag.so"
