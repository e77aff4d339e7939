#!/bin/bash
# Stamped dev build of the fp32 implicit-GEMM kernels (-DIGEMM_STAMPS) into build_dev/libscouter_igs.so; on the GPU box:
#   SCOUTER_HIP_LIB=build_dev/libscouter_igs.so python tools_dev/igemm_stamps.py
set -e
mkdir -p build_dev
python -c "from scouter_amd import _build; _build.build()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DIGEMM_STAMPS \
    -c scouter_amd/csrc/conv_igemm.hip -o build_dev/conv_igemm_igs.o 2>/dev/null
objs=$(ls scouter_amd/lib/obj/*.o | grep -v conv_igemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dev/libscouter_igs.so $objs build_dev/conv_igemm_igs.o
ls -la build_dev/libscouter_igs.so
