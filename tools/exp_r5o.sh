#!/bin/bash
# round 5: the last soak
cd $GRAFT_REPO_ROOT
timeout 200 python tools/stress.py 101 150 2>&1 | tail -2
HAP_AMD_PLACING_MIN_FRAMES=1 HAP_AMD_GRAPHS=1 timeout 200 python tools/stress.py 102 120 2>&1 | tail -2
HAP_AMD_FRAGMENT_INDEX=1 timeout 120 python tools/stress_threads.py 6 40 2>&1 | tail -1
timeout 150 python tools/fuzz_decode.py 5 2000 2>&1 | tail -1
timeout 150 python tools/fuzz_decode.py 6 800 --large 2>&1 | tail -1
timeout 150 python tools/fuzz_encode.py 7 2>&1 | tail -1
timeout 150 python tools/fuzz_blocks.py 8 2>&1 | tail -1
