#!/bin/bash
# builds glass-text-spotting_amd/libglass_hip_abl.so = the product library with ONE source recompiled with extra -D flags
# (timing-ablation instantiations): scripts/build_abl_lib.sh conv_h16.hip -DGLASS_H16_ABLATIONS ; use with GLASS_HIP_LIB=...
set -e
cd "$(dirname "$0")/../glass-text-spotting_amd"
python -c "import sys; sys.path.insert(0, '.'); from glass_amd import _lib; _lib.build_library()"
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $(python -c "import sys; sys.path.insert(0, '.'); from glass_amd import _lib; print(' '.join(_lib.DEVICE_FLAGS))") -I ../include -I csrc "$@" -c csrc/$src -o build/abl_$src.o
objs=$(ls build/*.hip.o | grep -v "build/abl_" | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libglass_hip_abl.so $objs build/abl_$src.o
echo built libglass_hip_abl.so
