export TMPDIR=/tmp
run() { # tag, env..., -- args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --queries 125000 --no-pmc --no-cpu-baseline --no-text-leg --verify 2000 --model-queries 0 --steps 24 "$@" > gpurun_out/r5_125k_$tag.json 2> gpurun_out/r5_125k_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_125k_$tag.err | sed 's/.*host thread): //' | cut -c1-150)"
}
run dt2 FMX_TURN_MIN=1000 FMX_DEVICE_TURNS=2 --
run dt1 FMX_TURN_MIN=1000 FMX_DEVICE_TURNS=1 --
run ht0 FMX_TURN_MIN_HOST=100000000 --
run ht0c6 FMX_TURN_MIN_HOST=100000000 FMX_MAX_HOST_ACTIVE=6 -- --text-calls 6 --host-threads 6
run c6 FMX_MAX_HOST_ACTIVE=6 -- --text-calls 6 --host-threads 6
run q12c6 GPU_MAX_HW_QUEUES=12 FMX_MAX_HOST_ACTIVE=6 -- --text-calls 6 --host-threads 6 --threads 10
