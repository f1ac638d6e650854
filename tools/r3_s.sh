export TMPDIR=/tmp
mkdir -p gpurun_out/r3s
python -m pytest tests -m gpu -x -q > gpurun_out/r3s/tests.log 2>&1; tail -3 gpurun_out/r3s/tests.log
bash tools/ab.sh r3s_c3 "" main
bash tools/ab.sh r3s_c2 "--config C2" main
bash tools/ab.sh r3s_q125 "--queries 125000" main
bash tools/ab.sh r3s_c5 "--config C5" main
bash tools/ab.sh r3s_c4 "--config C4" main
