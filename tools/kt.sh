# usage: bash tools/kt.sh <tag> [ENV=VAL ...] [-- bench args]  — kernel trace of the device-resident bench steps, summary into gpurun_out/
export TMPDIR=/tmp
tag=$1; shift
args=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; args="$@"; break; fi
  export "$1"; shift
done
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o runc -- python bench.py --child --steps 4 --warmup 2 $args > gpurun_out/kt_$tag.log 2>&1
python tools/kernel_trace_summary.py $(ls /tmp/kt_$tag/*kernel_trace.csv /tmp/kt_$tag/*/*kernel_trace.csv 2>/dev/null | tail -1) > gpurun_out/kt_${tag}_summary.csv
echo "== $tag"; grep -v rocprim gpurun_out/kt_${tag}_summary.csv | head -${KT_LINES:-16}
