python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python tools/stress.py 11 60 2>&1 | tail -2
HAP_AMD_PLACING_MIN_FRAMES=1 timeout 200 python tools/stress.py 13 60 2>&1 | tail -2
HAP_AMD_PLACING_MIN_FRAMES=2 HAP_AMD_NO_FUSION=1 timeout 200 python tools/stress.py 14 40 2>&1 | tail -2
timeout 300 python tools/fuzz_decode.py 2000 2>&1 | tail -2
