#!/bin/bash
# Ablated variants of the persistent plane kernel (PP_ABLATE bits: 1 no DMA in the K loop, 2 no DMA wait / barrier, 4 no LDS
# fragment reads, 8 no epilogue) into build_dev/libscouter_pp<N>.so -- results are WRONG, only the timing means something.
# usage (here, CPU): bash tools_dev/pp_ablate.sh 1 2 4 7 15 ; then on the GPU box:
#   for n in 1 2 4 7 15; do SCOUTER_HIP_LIB=build_dev/libscouter_pp$n.so python tools_dev/pp_ablate_bench.py; done
set -e
mkdir -p build_dev
python -c "from scouter_amd import _build; _build.build()"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DPP_ABLATE=$n \
      -c scouter_amd/csrc/conv_planes.hip -o build_dev/conv_planes_pp$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  objs=$(ls scouter_amd/lib/obj/*.o | grep -v conv_planes.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dev/libscouter_pp$n.so $objs build_dev/conv_planes_pp$n.o
done
ls -la build_dev/*pp*.so
