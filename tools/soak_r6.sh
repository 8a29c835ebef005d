#!/bin/bash
# r06 soak on the final code: stress (random textures, checker-made frames through the block scan and the workgroup-per-block
# decoder), fuzzed frames (small, large, blocks), fuzzed encodes, threads
cd $GRAFT_REPO_ROOT
timeout 400 python tools/stress.py 61 240 2>&1 | tail -3
timeout 300 python tools/fuzz_decode.py 62 1500 2>&1 | tail -2
timeout 300 python tools/fuzz_decode.py 63 800 --large --blocks 2>&1 | tail -2
timeout 300 python tools/fuzz_decode.py 65 600 --large --blocks --guess 2>&1 | tail -2
timeout 200 python tools/fuzz_encode.py 64 2>&1 | tail -2
timeout 200 python tools/stress_threads.py 2>&1 | tail -2
