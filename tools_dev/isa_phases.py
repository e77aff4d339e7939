"""Instruction census of a kernel's prologue / MFMA section / epilogue from hipcc's assembly (fp32 MFMA shares the
vector lanes with VALU, so per-block VALU work outside the K loop is matrix time lost on short-K layers).
usage: python tools_dev/isa_phases.py scouter_amd/csrc/conv_igemm.hip <mangled-name-substring> [...]"""
import re, subprocess, sys
src = sys.argv[1]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-S",
                "--cuda-device-only", "-o", "/tmp/_isa.s", src], capture_output=True)
text = open("/tmp/_isa.s").read()
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", text, re.M)]
for i, (pos, name) in enumerate(starts):
    if not any(s in name for s in sys.argv[2:]):
        continue
    end = text.find(".end_amdhsa_kernel", pos)          # the descriptor follows the code
    end = text.rfind("s_endpgm", pos, end) if end > 0 else (starts[i + 1][0] if i + 1 < len(starts) else len(text))
    body = [l.strip() for l in text[pos:end].split("\n")]
    body = [l for l in body if l and not l.startswith((";", ".", "_Z")) and not l.endswith(":")]
    idx = [k for k, l in enumerate(body) if l.startswith("v_mfma")]
    if not idx:
        continue

    def count(ls):
        return (sum(1 for l in ls if l.startswith("v_") and not l.startswith("v_mfma")),
                sum(1 for l in ls if re.match(r"v_\w+_f64", l)), sum(1 for l in ls if l.startswith("s_")),
                sum(1 for l in ls if l.startswith("ds_")), sum(1 for l in ls if l.startswith(("buffer_", "global_", "flat_", "scratch_"))))
    print(name[:100])
    print("  prologue  VALU %4d (f64 %3d) SALU %4d DS %3d MEM %3d" % count(body[:idx[0]]))
    print("  MFMA part VALU %4d (f64 %3d) SALU %4d DS %3d MEM %3d  MFMA %d" % (count(body[idx[0]:idx[-1] + 1]) + (len(idx),)))
    print("  epilogue  VALU %4d (f64 %3d) SALU %4d DS %3d MEM %3d" % count(body[idx[-1] + 1:]))
