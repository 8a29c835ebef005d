#!/bin/bash
# Dev tool: builds timing-experiment variants of the library into gpurun-visible files
#   tools/build_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"-DFLAG3" ...
# -> hap_amd/variants/libhap_amd_<name>.so  (load with HAP_AMD_LIBRARY=...)
set -e
cd "$(dirname "$0")/../hap_amd/csrc"
mkdir -p ../variants
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-function -DHAP_MEASUREMENT_BUILD"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  make -s BUILD=build_$name OUT=../variants/libhap_amd_$name.so HIPFLAGS="$BASE $flags" >/dev/null
  echo built $name "($flags)"
done
