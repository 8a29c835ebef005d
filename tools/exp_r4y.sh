timeout 150 python tools/stress.py 71 60 2>&1 | tail -1
HAP_AMD_PLACING_MIN_FRAMES=1 HAP_AMD_GRAPHS=1 timeout 150 python tools/stress.py 72 50 2>&1 | tail -1
timeout 120 python tools/stress_threads.py 6 30 2>&1 | tail -1
timeout 120 python tools/fuzz_decode.py 1200 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
