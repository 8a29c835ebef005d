#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5i_pytest.log
HAP_AMD_PLACING_MIN_FRAMES=1 HAP_AMD_GRAPHS=1 timeout 300 python tools/stress.py 82 120 2>&1 | tail -3
HAP_AMD_PLACING_MIN_FRAMES=1 HAP_AMD_GRAPHS=1 timeout 200 python tools/stress.py 83 90 2>&1 | tail -3
HAP_AMD_GRAPHS=1 timeout 200 python tools/stress.py 84 60 2>&1 | tail -3
