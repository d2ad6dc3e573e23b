#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> paif_amd/lib/libpaif_hip_<name>.so (same ABI, for A/B runs via PAIF_LIB)
name=$1; shift
cd /root/repo
objs=""
mkdir -p /tmp/variant_$name
for f in paif_amd/csrc/*.hip; do
  o=/tmp/variant_$name/$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I paif_amd/csrc "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o paif_amd/lib/libpaif_hip_$name.so $objs && echo built paif_amd/lib/libpaif_hip_$name.so
