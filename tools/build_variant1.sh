#!/bin/bash
# tools/build_variant1.sh <name> <file.hip> <extra hipcc flags...> -> paif_amd/lib/libpaif_hip_<name>.so with ONE source recompiled
# (the other objects are the current build's, paif_amd/lib/obj -- run __graft_entry__.build() first): quick A/B builds of one kernel file
name=$1; file=$2; shift; shift
cd /root/repo
rm -rf /tmp/variant_$name; mkdir -p /tmp/variant_$name
cp paif_amd/lib/obj/*.o /tmp/variant_$name/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I paif_amd/csrc "$@" -c paif_amd/csrc/$file -o /tmp/variant_$name/$(basename $file .hip).o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o paif_amd/lib/libpaif_hip_$name.so /tmp/variant_$name/*.o && echo built paif_amd/lib/libpaif_hip_$name.so
