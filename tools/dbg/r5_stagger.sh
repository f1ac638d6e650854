# design-time: the value leg (20 steps from an idle pipeline), 4..6 text calls in flight, with the library's pacing of call starts
# (bench.py keeps Python's garbage collector out of its timed regions since this script's second run)
export TMPDIR=/tmp
run() { tag=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 20 --warmup 5 "$@" > gpurun_out/r5_st_$tag.json 2> gpurun_out/r5_st_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_st_$tag.err | sed 's/.*host thread): //' | cut -c1-170) | $(grep -h 'synchronous callers' gpurun_out/r5_st_$tag.err | sed 's/.*callers: //' | cut -c1-60)"; }
run p4a --
run p5a -- --text-calls 5
run p4b --
run p5b -- --text-calls 5
run p6a -- --text-calls 6
run p5c -- --text-calls 5
