# design-time: the value leg (20 steps from an idle pipeline), 4..6 text calls in flight, with the library's pacing of call starts
export TMPDIR=/tmp
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 20 --warmup 5 "$@" > gpurun_out/r5_st_$tag.json 2> gpurun_out/r5_st_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_st_$tag.err | sed 's/.*host thread): //' | cut -c1-120) | $(grep -h 'synchronous callers' gpurun_out/r5_st_$tag.err | sed 's/.*callers: //' | cut -c1-60)"; }
run p4 --
run p5a FMX_MAX_HOST_ACTIVE=5 -- --text-calls 5 --host-threads 5
run p5b FMX_MAX_HOST_ACTIVE=5 -- --text-calls 5 --host-threads 5
run p5c FMX_MAX_HOST_ACTIVE=5 -- --text-calls 5 --host-threads 5
run p5off FMX_MAX_HOST_ACTIVE=5 FMX_HOST_SPACING_US=0 -- --text-calls 5 --host-threads 5
run p6 FMX_MAX_HOST_ACTIVE=6 -- --text-calls 6 --host-threads 6
