# usage: bash tools/kt_default.sh <tag> [ENV=VAL ...] [-- leg args] — kernel trace of the default-index leg (tools/default_index_leg.py), summary into gpurun_out/
export TMPDIR=/tmp
tag=$1; shift
args="--config C3 --steps 6 --verify 0"
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; args="$@"; break; fi
  export "$1"; shift
done
rm -rf /tmp/ktd_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktd_$tag -o runc -- python tools/default_index_leg.py $args > gpurun_out/ktd_$tag.log 2>&1
# (the leg's fmx-cli children are traced too: the leg's own trace is the largest file)
python tools/kernel_trace_summary.py $(ls -S /tmp/ktd_$tag/*kernel_trace.csv /tmp/ktd_$tag/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/ktd_${tag}_summary.csv
echo "== $tag"; grep -v rocprim gpurun_out/ktd_${tag}_summary.csv | head -${KT_LINES:-24}
