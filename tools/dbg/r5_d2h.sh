# design-time: priority of the stream the records go home on (a copy kernel on this platform)
export TMPDIR=/tmp
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs FMX_LIB=$PWD/build/libfmx_d2hprio.so python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 20 --warmup 5 "$@" > gpurun_out/r5_d2h_$tag.json 2> gpurun_out/r5_d2h_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_d2h_$tag.err | sed 's/.*host thread): //' | cut -c1-170) | $(grep -h 'synchronous callers' gpurun_out/r5_d2h_$tag.err | sed 's/.*callers: //' | cut -c1-60)"; }
run lo0 FMX_D2H_PRIO=0 --
run hi0 FMX_D2H_PRIO=2 --
run lo1 FMX_D2H_PRIO=0 --
run hi1 FMX_D2H_PRIO=2 --
run mid FMX_D2H_PRIO=1 --
