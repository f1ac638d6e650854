# design-time: runtime copy-engine settings against the value leg
export TMPDIR=/tmp
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 20 --warmup 5 "$@" > gpurun_out/r5_hsa_$tag.json 2> gpurun_out/r5_hsa_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_hsa_$tag.err | sed 's/.*host thread): //' | cut -c1-170)"; }
run base --
run nosdma HSA_ENABLE_SDMA=0 --
run gang HSA_ENABLE_SDMA_GANG=1 --
run base2 --
