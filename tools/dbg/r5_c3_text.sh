export TMPDIR=/tmp
run() { # tag, env..., -- args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --no-text-leg --verify 2000 --model-queries 0 --steps 16 "$@" > gpurun_out/r5_c3t_$tag.json 2> gpurun_out/r5_c3t_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_c3t_$tag.err | sed 's/.*host thread): //' | cut -c1-150)"
  echo "    $(grep -h 'synchronous callers' gpurun_out/r5_c3t_$tag.err | sed 's/.*callers: //')"
}
run base --
run ht0 FMX_TURN_MIN_HOST=100000000 --
run ht1 FMX_TURNS=1 --
