timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for cfg in C2 C1; do
  timeout 200 python tools/probe_placed.py $cfg 60 5 2>&1 | tail -1
  HAP_AMD_NO_FUSION=1 timeout 200 python tools/probe_placed.py $cfg 60 5 2>&1 | tail -1
done
timeout 200 python tools/stress.py 21 40 2>&1 | tail -2
