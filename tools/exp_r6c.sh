#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_variants.sh C5 4 one two one two 2>&1 | grep "^==\|^fields"
bash tools/ab_variants.sh C4 30 one two one two 2>&1 | grep "^==\|^fields"
