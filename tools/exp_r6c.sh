#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/probe_plain.py C4 1 2>&1 | grep plain
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
