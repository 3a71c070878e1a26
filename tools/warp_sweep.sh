for tx in 8 16 32 64; do for nt in 0 1; do echo "TX=$tx NT=$nt"; VSTAB_WARP_TX=$tx VSTAB_WARP_NT=$nt python tools/warp_microbench.py --n 256 --reps 6 2>&1 | tail -1; done; done
