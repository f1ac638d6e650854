# design-time: how the runtime moves the records home (blit kernel / SDMA) and what it costs the value leg
export TMPDIR=/tmp
run() { # tag, env..., -- args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 20 --warmup 5 "$@" > gpurun_out/r5_blit_$tag.json 2> gpurun_out/r5_blit_$tag.err
  echo "$tag: $(grep -h 'value (' gpurun_out/r5_blit_$tag.err | sed 's/.*host thread): //' | cut -c1-190)"
}
run base --
run noexp FMX_EXPORT_KERNEL=0 --
run base5 FMX_MAX_HOST_ACTIVE=5 -- --text-calls 5 --host-threads 5
run noexp5 FMX_EXPORT_KERNEL=0 FMX_MAX_HOST_ACTIVE=5 -- --text-calls 5 --host-threads 5
rm -rf /tmp/tt
FMX_EXPORT_KERNEL=0 FMX_CHILD_MODE=text rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tt -o run -- python bench.py --child --steps 8 --warmup 4 > gpurun_out/tt.log 2>&1
python tools/dbg/timeline.py /tmp/tt 6 40 > gpurun_out/tt_timeline_noexp.txt
grep -c copyBuffer gpurun_out/tt_timeline_noexp.txt; grep -c "DEVICE_TO_HOST" gpurun_out/tt_timeline_noexp.txt
