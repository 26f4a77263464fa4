"""kernel resource usage of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, scratch, occupancy, LDS.
usage: python tools/r5/resusage.py csrc/pwfused.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/resusage.o"] + sys.argv[2:]
p = subprocess.run(cmd, capture_output=True, text=True)
if p.returncode:
    print(p.stderr[-4000:])
    sys.exit(1)
blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
for b in blocks:
    name = b.split("\n")[0].strip().split()[0]
    dn = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "").replace("void ", "")
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print("%-64s vgpr %4s agpr %4s sgpr %3s scratch %4s occ %s lds %6s" % (
        dn[:64], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]")))
