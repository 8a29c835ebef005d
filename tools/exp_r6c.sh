#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in 1 2 3 4; do timeout 300 python tools/probe_foreign.py $n 7680 4320 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "workgroup or scan or reference" 2>&1 | tail -2
