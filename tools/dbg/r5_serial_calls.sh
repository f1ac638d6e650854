# design-time: text calls in flight with ONE compute stream per call (FMX_SERIAL), unchained (FMX_TURN_MIN_HOST above any batch), paced starts
export TMPDIR=/tmp
run() { # tag, env..., -- args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 20 --warmup 5 "$@" > gpurun_out/r5_sc_$tag.json 2> gpurun_out/r5_sc_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_sc_$tag.err | sed 's/.*host thread): //' | cut -c1-200)"
}
U=FMX_TURN_MIN_HOST=1000000000
run s5u FMX_SERIAL=1 $U -- --text-calls 5
run s6u FMX_SERIAL=1 $U -- --text-calls 6
run s7u FMX_SERIAL=1 $U -- --text-calls 7
run p4 -- --text-calls 4
