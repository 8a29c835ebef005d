#!/bin/bash
# the round's last soak, on the final code
cd $GRAFT_REPO_ROOT
timeout 130 python tools/stress.py 401 120 2>&1 | tail -1
HAP_AMD_PLACING_MIN_FRAMES=1 HAP_AMD_GRAPHS=1 timeout 100 python tools/stress.py 402 80 2>&1 | tail -1
HAP_AMD_FRAGMENT_INDEX=1 timeout 100 python tools/stress_threads.py 6 30 2>&1 | tail -1
timeout 120 python tools/fuzz_decode.py 21 1200 --guess 2>&1 | tail -1
timeout 100 python tools/fuzz_decode.py 22 1200 2>&1 | tail -1
timeout 100 python tools/fuzz_decode.py 23 500 --large --blocks 2>&1 | tail -1
timeout 100 python tools/fuzz_encode.py 24 2>&1 | tail -1
