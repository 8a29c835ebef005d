echo "== unfused, placed"; HAP_AMD_NO_FUSION=1 timeout 200 python tools/probe_placed.py C4 60 5 2>&1 | tail -1
echo "== unfused, no placing"; HAP_AMD_NO_FUSION=1 HAP_AMD_NO_PLACING=1 timeout 200 python tools/probe_placed.py C4 60 5 2>&1 | tail -1
