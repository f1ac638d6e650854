# usage: bash tools/kt.sh <tag> [ENV=VAL ...]  — kernel trace of the device-resident bench steps, summary into gpurun_out/
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$tag -o runc -- python bench.py --child --steps 4 --warmup 2 > gpurun_out/kt_$tag.log 2>&1
python tools/kernel_trace_summary.py $(ls /tmp/kt_$tag/*kernel_trace.csv /tmp/kt_$tag/*/*kernel_trace.csv 2>/dev/null | tail -1) > gpurun_out/kt_${tag}_summary.csv
echo "== $tag $@"; grep -v rocprim gpurun_out/kt_${tag}_summary.csv | head -${KT_LINES:-14}
