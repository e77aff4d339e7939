#!/bin/bash
# Builds ablated variants of the resident-rows plane kernel (PHALO_ABLATE bits: 1 no DMA in the K loop, 2 no DMA wait /
# barrier, 4 no LDS fragment reads) into build_dev/libscouter_ab<N>.so -- results are WRONG, only the timing means
# something.  usage (here, CPU): bash tools_dev/phalo_ablate.sh 1 2 4 7 ; then on the GPU box:
#   SCOUTER_HIP_LIB=build_dev/libscouter_ab1.so python tools_dev/planes_bench.py
set -e
mkdir -p build_dev
python -c "from scouter_amd import _build; _build.build()"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DPHALO_ABLATE=$n \
      -c scouter_amd/csrc/conv_planes.hip -o build_dev/conv_planes_ab$n.o &
done
wait
for n in "$@"; do
  objs=$(ls scouter_amd/lib/obj/*.o | grep -v conv_planes.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dev/libscouter_ab$n.so $objs build_dev/conv_planes_ab$n.o
done
ls -la build_dev/*.so
