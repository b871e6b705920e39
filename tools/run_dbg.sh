for d in 0 1 2 3 7; do echo "== dbg=$d"; TMDIFF_CONV_DEBUG=$d timeout -k 10 100 python tools/bench_conv.py 32 3 2>&1 | grep -E "L0 32->32 k3|L1 128->128|L3 256->256|L3 768->128 k3|weighted"; done
