FMX_BENCH_ORDER=host_first bash tools/ab.sh r3u_c2 "--config C2" main
FMX_BENCH_ORDER=host_first FMX_TURNS=1 bash tools/ab.sh r3u_c2t1 "--config C2" main
FMX_BENCH_ORDER=host_first bash tools/ab.sh r3u_c3 "" main
FMX_BENCH_ORDER=host_first FMX_TURNS=1 bash tools/ab.sh r3u_c3t1 "" main
