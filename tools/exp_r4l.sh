for cfg_nf in "C4 1" "C4 2" "C4 4" "C4 8" "C4 16" "C2 60" "C3 60" "C5 4" "C1 1" "C1 60"; do
  set -- $cfg_nf
  echo "== $1 $2 placed";   timeout 200 python tools/probe_placed.py $1 $2 5 2>&1 | tail -1
  echo "== $1 $2 gathered"; HAP_AMD_NO_PLACING=1 timeout 200 python tools/probe_placed.py $1 $2 5 2>&1 | tail -1
done
