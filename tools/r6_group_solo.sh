for k in 1 2 3 4 5 6 7 8; do GKRHIP_BENCH_OPTIONS=g_max=13 timeout 300 python tools/r6_group_probe.py 20 $k $k 3 2>&1 | grep "in flight"; done
