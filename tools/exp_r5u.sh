#!/bin/bash
# soak after the plain-frame work: stress (plain frames with the pre-pass forced in a third of the rounds), fuzzed table-less
# frames through the forced pre-pass, the usual fuzzers, the decode-to-pictures tests
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rgba or pictures or padded" 2>&1 | tail -2
timeout 250 python tools/stress.py 201 200 2>&1 | tail -2
HAP_AMD_PLACING_MIN_FRAMES=1 HAP_AMD_GRAPHS=1 timeout 200 python tools/stress.py 202 120 2>&1 | tail -2
timeout 250 python tools/fuzz_decode.py 11 1500 --guess 2>&1 | tail -3
timeout 150 python tools/fuzz_decode.py 12 1500 2>&1 | tail -1
timeout 150 python tools/fuzz_decode.py 13 600 --large --blocks 2>&1 | tail -1
