for a in 0 1 3 7 8; do echo "ablate=$a"; VSTAB_DIS_ABLATE=$a python bench.py --steps 3 --warmup 1 --cpu-frames 0 2>&1 | tail -1 | grep -o '"dis": [0-9.]*'; done
