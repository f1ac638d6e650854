for th in 4 6; do
FMX_BENCH_THREADS=$th bash tools/ab.sh r3t_c2_$th "--config C2" main
FMX_BENCH_THREADS=$th bash tools/ab.sh r3t_q125_$th "--queries 125000" main
done
FMX_BENCH_THREADS=2 bash tools/ab.sh r3t_c3_2 "" main
FMX_BENCH_THREADS=4 bash tools/ab.sh r3t_c3_4 "" main
