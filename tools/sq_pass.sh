# usage: bash tools/sq_pass.sh <tag> [ENV=VAL ...] [-- bench args]  — SQ counters of the staged device-resident steps (2 timed steps), per kernel
export TMPDIR=/tmp
tag=$1; shift
args=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; args="$@"; break; fi
  export "$1"; shift
done
rm -rf /tmp/sq_$tag
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/sq_$tag -o run -- python bench.py --child --steps 2 --warmup 1 $args > gpurun_out/sq_$tag.log 2>&1
f=$(ls /tmp/sq_$tag/*counter_collection.csv /tmp/sq_$tag/*/*counter_collection.csv 2>/dev/null | tail -1)
python tools/pmc_summary.py $f 2 > gpurun_out/sq_${tag}.csv
echo "== $tag"; grep -E "expand_lf|expand_b<true>|expand_b<false>" gpurun_out/sq_${tag}.csv
