"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel (registers, scratch, occupancy, LDS).
usage: transformergrooveinfilling_amd/csrc/build.sh -Rpass-analysis=kernel-resource-usage 2> res.txt; python tools/kernel_resources.py res.txt [filter]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
names = [b.split(" [-Rpass")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for b, nm in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    nm = re.sub(r"\(.*\)$", "", nm.replace("void ", ""))
    if flt and flt not in nm and not (flt == "scratch" and g(r"ScratchSize \[bytes/lane\]") > 0):
        continue
    print("%-72s VGPR %3d AGPR %3d scratch %4d B/lane  waves/SIMD %d  LDS %6d B  SGPR %3d" %
          (nm[:72], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g("SGPRs")))
