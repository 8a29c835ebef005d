for round in 1 2; do
for v in head new; do
  echo "== $v"; HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_$v.so timeout 200 python tools/probe_placed.py C3 60 7 2>&1 | tail -1
done; done
timeout 600 python -m pytest tests -x -q -m gpu -k "placed or fused or row_pitch or round_trip or full_size" 2>&1 | tail -2
