#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/probe_plain.py C4 1 2>&1 | grep plain
python tools/probe_plain.py C1 1 2>&1 | grep plain
timeout 300 python tools/probe_foreign.py 2 7680 4320 2>&1 | tail -3 | grep "block scan"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "workgroup or scan or reference or table_less or plain or malformed or corrupt or pieces" 2>&1 | tail -2
