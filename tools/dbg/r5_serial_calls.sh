# design-time: text calls in flight with one compute stream per call (FMX_SERIAL) against the default pair
export TMPDIR=/tmp
run() { # tag, env..., -- args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 16 "$@" > gpurun_out/r5_sc_$tag.json 2> gpurun_out/r5_sc_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_sc_$tag.err | sed 's/.*host thread): //' | cut -c1-200)"
}
run base FMX_PROBE_SPLIT=0 --
run s4 FMX_PROBE_SPLIT=0 FMX_SERIAL=1 --
run s6 FMX_PROBE_SPLIT=0 FMX_SERIAL=1 FMX_MAX_HOST_ACTIVE=6 -- --text-calls 6 --host-threads 6 --threads 8
run s8 FMX_PROBE_SPLIT=0 FMX_SERIAL=1 FMX_MAX_HOST_ACTIVE=8 -- --text-calls 8 --host-threads 8 --threads 8
run s10 FMX_PROBE_SPLIT=0 FMX_SERIAL=1 FMX_MAX_HOST_ACTIVE=10 -- --text-calls 10 --host-threads 10 --threads 10
