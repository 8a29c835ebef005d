#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for cfg in "C5 4" "C4 60"; do set -- $cfg; python tools/probe_fields.py $1 $2 2>&1 | grep "^fields\|PARITY"; done
done
timeout 600 python -m pytest tests -m gpu -x -q -k "field or stream or table or decode" 2>&1 | tail -2
