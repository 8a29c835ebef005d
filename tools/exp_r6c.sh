#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/fuzz_decode.py 91 600 --large --blocks --guess 2>&1 | tail -1
python tools/probe_plain.py C4 1 2>&1 | grep plain
