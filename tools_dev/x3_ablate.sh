#!/bin/bash
# Ablated variants of the register-split GEMM (X3_ABLATE bits, csrc/conv_x3.hip) into build_dev/libscouter_x3a<N>.so -- results
# are WRONG, only the timing means something.  usage (here, CPU): bash tools_dev/x3_ablate.sh 1 2 4 6 8 16 ... ; on the GPU box:
#   for n in ...; do SCOUTER_HIP_LIB=build_dev/libscouter_${T}$n.so python tools_dev/x3_ablate_bench.py; done
set -e
mkdir -p build_dev
python -c "from scouter_amd import _build; _build.build()"
# XW=1 bash tools_dev/x3_ablate.sh ... : the same for the weight-gradient kernel (XW_ABLATE bits) into libscouter_xwa<N>.so
D=X3_ABLATE; T=x3a
if [ "${XW:-0}" = "1" ]; then D=XW_ABLATE; T=xwa; fi
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -D$D=$n \
      -c scouter_amd/csrc/conv_x3.hip -o build_dev/conv_${T}_$n.o 2>/dev/null &
done
wait
for n in "$@"; do
  objs=$(ls scouter_amd/lib/obj/*.o | grep -v conv_x3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dev/libscouter_${T}$n.so $objs build_dev/conv_${T}_$n.o
done
ls build_dev/*${T}*.so | wc -l
