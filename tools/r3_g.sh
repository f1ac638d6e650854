export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
python -m pytest tests/test_facade.py -x -q > gpurun_out/r3g/facade.log 2>&1; tail -3 gpurun_out/r3g/facade.log
printf 'x\t0.5\t0\t2\taa bb\taa bb\t\n' > /tmp/cases.tsv
tests/cpp/facade_test tests/golden/tm1.fmi /tmp/cases.tsv > gpurun_out/r3g/facade_out.txt 2> gpurun_out/r3g/facade_err.txt; echo "rc=$?"
tail -5 gpurun_out/r3g/facade_out.txt
