# usage: bash tools/sq_default.sh <tag> [ENV=VAL ...] [-- leg args] — SQ counters of the default-index leg's kernels (tools/default_index_leg.py)
export TMPDIR=/tmp
tag=$1; shift
args="--config C3 --steps 2 --warmup 1 --verify 0"
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; args="$@"; break; fi
  export "$1"; shift
done
rm -rf /tmp/sqd_$tag
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/sqd_$tag -o run -- python tools/default_index_leg.py $args > gpurun_out/sqd_$tag.log 2>&1
f=$(ls -S /tmp/sqd_$tag/*counter_collection.csv /tmp/sqd_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
python tools/pmc_summary.py $f 2 > gpurun_out/sqd_${tag}.csv
echo "== $tag"; grep -E "kernel|k_dp" gpurun_out/sqd_${tag}.csv
