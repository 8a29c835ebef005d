#!/bin/bash
# round 5: soak -- stress in three modes, threads, fuzzed frames (small and large), smoke
cd $GRAFT_REPO_ROOT
timeout 200 python tools/stress.py 81 90 2>&1 | tail -3
HAP_AMD_PLACING_MIN_FRAMES=1 HAP_AMD_GRAPHS=1 timeout 150 python tools/stress.py 82 60 2>&1 | tail -3
timeout 120 python tools/stress_threads.py 6 30 2>&1 | tail -2
timeout 150 python tools/fuzz_decode.py 3 1500 2>&1 | tail -2
timeout 150 python tools/fuzz_decode.py 4 600 --large 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
