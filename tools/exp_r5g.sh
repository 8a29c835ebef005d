#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5g_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5g_pytest.log
bash tools/evidence_round.sh
