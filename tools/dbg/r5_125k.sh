export TMPDIR=/tmp
run() { # tag, env..., -- args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --queries 125000 --no-pmc --no-cpu-baseline --no-text-leg --verify 2000 --model-queries 0 --steps 24 "$@" > gpurun_out/r5_125k_$tag.json 2> gpurun_out/r5_125k_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_125k_$tag.err | sed 's/.*host thread): //' | cut -c1-150)"
}
run base -- 
run pb256 FMX_LIB=$PWD/build/libfmx_pb256.so --
run pb512 FMX_LIB=$PWD/build/libfmx_pb512.so --
run t8 -- --threads 8
run t10 -- --threads 10 
run hg8k FMX_HEAVY_GRID=8192 FMX_B_GRID=2048 --
run q12 GPU_MAX_HW_QUEUES=12 -- --threads 10
