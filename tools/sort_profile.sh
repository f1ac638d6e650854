# kernel trace of the suffix sort at 5 M sentences: bash tools/sort_profile.sh [n_sentences]
export TMPDIR=/tmp
n=${1:-5000000}
out=gpurun_out/prof_sort
rm -rf $out; mkdir -p $out
FMX_SORT_TIMING=1 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o run -- python tools/sort_timing.py --child $n /tmp/sort_prof.npy > $out/kt.log 2>&1
python tools/kernel_trace_summary.py $(find $out/kt -name '*kernel_trace.csv' | head -1) > $out/sort_kernel_trace_summary.csv 2>>$out/kt.log
find $out -name '*kernel_trace.csv' -delete
grep "fmx sort" $out/kt.log
head -25 $out/sort_kernel_trace_summary.csv | cut -c1-160
