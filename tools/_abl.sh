timeout -k 10 300 python -m pytest tests/test_message_tile_gpu.py -m gpu -x -q 2>&1 | tail -5
for d in 0 1 2; do echo "dbg=$d"; MPNN_MT_DEBUG=$d timeout -k 10 100 python tools/bench_message_tile.py 2>&1 | grep "tile kernel\|max"; done
