# design-time: the value leg with text calls chained at the predecessor's mid-point (default) / unchained, 5 and 6 in flight
export TMPDIR=/tmp
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 20 --warmup 5 "$@" > gpurun_out/r5_uc_$tag.json 2> gpurun_out/r5_uc_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_uc_$tag.err | sed 's/.*host thread): //' | cut -c1-120)"; }
U=FMX_TURN_MIN_HOST=1000000000
run c5 --
run u5 $U --
run u6 $U -- --text-calls 6
run c5b --
run u5b $U --
run t1 FMX_TURNS=1 --
