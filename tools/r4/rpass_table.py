"""Summarise a hipcc -Rpass-analysis=kernel-resource-usage log: one line per kernel (VGPRs, AGPRs, scratch bytes per lane,
waves per SIMD, LDS bytes).  usage: python tools/r4/rpass_table.py LOG [substring ...]"""
import re
import subprocess
import sys


def main():
    t = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    rows = []
    for b in re.split(r"(?=remark: Function Name)", t):
        m = re.search(r"Function Name: (\S+)", b)
        if not m:
            continue

        def g(k):
            mm = re.search(re.escape(k) + r": (\d+)", b)
            return int(mm.group(1)) if mm else -1
        rows.append((m.group(1), g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
                     g("LDS Size [bytes/block]")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print("%5s %5s %7s %4s %6s  kernel" % ("VGPR", "AGPR", "scratch", "occ", "LDS"))
    for (n, v, a, sc, occ, lds), d in zip(rows, names):
        d = d.replace("(anonymous namespace)::", "").replace("void ", "")
        d = re.sub(r"\(.*", "", d)
        if pats and not all(p in d for p in pats):
            continue
        print("%5d %5d %7d %4d %6d  %s" % (v, a, sc, occ, lds, d))


if __name__ == "__main__":
    main()
