for t in 1 2; do FMX_TURNS=$t bash tools/ab.sh r3n_t$t "" main; done
FMX_TURNS=2 bash tools/ab.sh r3n_c2 "--config C2" main
FMX_TURNS=2 bash tools/ab.sh r3n_q125 "--queries 125000" main
FMX_TURNS=1 bash tools/ab.sh r3n_q125t1 "--queries 125000" main
