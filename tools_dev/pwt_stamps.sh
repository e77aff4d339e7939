#!/bin/bash
# Stamped dev build of the tap-fused plane weight gradient (-DPWT_STAMPS) into build_dev/libscouter_pwt.so; on the GPU box:
#   SCOUTER_HIP_LIB=build_dev/libscouter_pwt.so python tools_dev/pwt_stamps.py
set -e
mkdir -p build_dev
python -c "from scouter_amd import _build; _build.build()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DPWT_STAMPS ${PWT_EXTRA:-} \
    -c scouter_amd/csrc/conv_planes.hip -o build_dev/conv_planes_pwt.o 2>/dev/null
objs=$(ls scouter_amd/lib/obj/*.o | grep -v conv_planes.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dev/libscouter_pwt.so $objs build_dev/conv_planes_pwt.o
ls -la build_dev/libscouter_pwt.so
