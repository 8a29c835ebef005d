for v in cur abl1 abl2 abl3; do
  lib=$PWD/hap_amd/variants/libhap_amd_$v.so
  [ "$v" = cur ] && lib=$PWD/hap_amd/libhap_amd.so
  echo "== $v"; HAP_AMD_LIBRARY=$lib timeout 200 python tools/probe_placed.py C4 60 5 2>&1 | tail -1
done
echo "== no placing"; HAP_AMD_NO_PLACING=1 timeout 200 python tools/probe_placed.py C4 60 5 2>&1 | tail -1
