"""Tabulates hipcc's -Rpass-analysis=kernel-resource-usage remarks (VGPR / AGPR / SGPR / LDS / occupancy per kernel).
usage: python tools_dev/kernel_resources.py scouter_amd/csrc/conv_igemm.hip [name-substring]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c", src,
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("%-70s %5s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
for c in rows:
    n = re.sub(r"\(.*", "", c["name"]).replace("void ", "")
    if flt in n:
        print("%-70s %5s %5s %5s %8s %4s %7s" % (n[:70], c.get("VGPRs"), c.get("AGPRs"), c.get("TotalSGPRs"),
                                                 c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"),
                                                 c.get("LDS Size [bytes/block]")))
