#!/bin/bash
# builds glass-text-spotting_amd/libglass_hip_<name>.so = every csrc/*.hip compiled with extra flags (kernel experiments;
# select it with GLASS_HIP_LIB=...):  scripts/build_variant_lib.sh nopk -Xclang -target-feature -Xclang -packed-fp32-ops
set -e
cd "$(dirname "$0")/../glass-text-spotting_amd"
name=$1; shift
mkdir -p build/variant_$name
pids=()
for s in csrc/*.hip; do
  o=build/variant_$name/$(basename $s).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../include -I csrc "$@" -c $s -o $o 2> build/variant_$name/$(basename $s).log &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libglass_hip_$name.so build/variant_$name/*.o
echo built libglass_hip_$name.so
