python tools/probe_holdoff.py fff 24 2>/dev/null | awk '{printf "%s:%s ", $1, $3}'; echo
HAP_AMD_MEMSET_NODES=1 python tools/probe_holdoff.py fff 24 2>/dev/null | awk '{printf "%s:%s ", $1, $3}'; echo
