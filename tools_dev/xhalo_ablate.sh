#!/bin/bash
# Ablated variants of the resident-rows register-split kernel (XH_ABLATE bits, csrc/conv_xhalo.hip) into
# build_dev/libscouter_xh<N>.so -- results are WRONG, only the timing means something.  usage (here, CPU):
#   bash tools_dev/xhalo_ablate.sh 1 2 4 8 ...     then on the GPU box:
#   for n in ...; do SCOUTER_HIP_LIB=build_dev/libscouter_xh$n.so python tools_dev/xhalo_time.py; done
set -e
mkdir -p build_dev
python -c "from scouter_amd import _build; _build.build()"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -DXH_ABLATE=$n \
      -c scouter_amd/csrc/conv_xhalo.hip -o build_dev/conv_xh_$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  objs=$(ls scouter_amd/lib/obj/*.o | grep -v conv_xhalo.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dev/libscouter_xh$n.so $objs build_dev/conv_xh_$n.o
done
ls build_dev/*xh*.so | wc -l
