#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$PWD/hap_amd/variants/libhap_amd_m.so
for c in C4 C1; do
echo "== $c half segments"; HAP_AMD_LIBRARY=$L python tools/probe_plain.py $c 1 2>&1 | grep plain
echo "== $c full segments"; HAP_AMD_SCAN_SEGMENT_FULL=1 HAP_AMD_LIBRARY=$L python tools/probe_plain.py $c 1 2>&1 | grep plain
done
echo "== reference frame, full segments"; HAP_AMD_SCAN_SEGMENT_FULL=1 HAP_AMD_LIBRARY=$L python tools/probe_foreign.py 1 7680 4320 2>&1 | tail -1
echo "== reference frame, half"; HAP_AMD_LIBRARY=$L python tools/probe_foreign.py 1 7680 4320 2>&1 | tail -1
