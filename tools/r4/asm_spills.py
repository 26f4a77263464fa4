"""Where does a kernel spill?  For one kernel of a hipcc -S dump: scratch loads/stores per basic block, with the block's
MFMA count (K loop blocks have MFMAs).  usage: python tools/r4/asm_spills.py FILE.s 'demangled substring'"""
import re
import subprocess
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    for (i, n), d in zip(starts, names):
        if pat not in d:
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith("\t.section") and not re.match(r"^\.Lfunc_end", lines[j]):
            j += 1
        body = lines[i:j]
        print(d[:160], "lines", len(body))
        blk, stats = "entry", {}
        order = []
        for l in body:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                blk = m.group(1)
            s = stats.setdefault(blk, [0, 0, 0, 0])
            if blk not in order:
                order.append(blk)
            if "scratch_store" in l:
                s[0] += 1
            if "scratch_load" in l:
                s[1] += 1
            if "v_mfma" in l:
                s[2] += 1
            s[3] += 1
        for b in order:
            st, ld, mf, n_ = stats[b]
            if st or ld or mf:
                print("  %-12s insts %5d  mfma %4d  scratch_store %3d  scratch_load %3d" % (b, n_, mf, st, ld))


if __name__ == "__main__":
    main()
