for t in 1 2; do FMX_TURNS=$t bash tools/ab.sh r3r_t$t "" main; done
FMX_TURNS=2 bash tools/ab.sh r3r_c2 "--config C2" main
FMX_TURNS=1 bash tools/ab.sh r3r_c2t1 "--config C2" main
FMX_TURNS=0 bash tools/ab.sh r3r_c2t0 "--config C2" main
FMX_TURNS=2 bash tools/ab.sh r3r_q125 "--queries 125000" main
FMX_TURNS=0 bash tools/ab.sh r3r_q125t0 "--queries 125000" main
