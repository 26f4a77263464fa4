"""per kernel: the basic blocks that hold both global loads and stores, with the order of their memory operations and
vmcnt waits (round 5: are the stores counted, or does the block end in vmcnt(0)?).
usage: python tools/r5/asm_loop_waits.py file.s <kernel substring> [min loads] [min stores]"""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mins = int(sys.argv[4]) if len(sys.argv) > 4 else 2
st = next(i for i, l in enumerate(L) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
en = next(i for i in range(st, len(L)) if L[i].startswith(".Lfunc_end"))
cur, blocks = None, []
for l in L[st:en]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
    elif cur:
        cur[1].append(l)
for name, b in blocks:
    nl, ns = sum("global_load" in x for x in b), sum("global_store" in x for x in b)
    if nl >= minl and ns >= mins:
        seq = []
        for x in b:
            if "global_load" in x:
                seq.append("L")
            elif "global_store" in x:
                seq.append("S")
            elif "scratch_" in x:
                seq.append("X")
            else:
                m = re.search(r"vmcnt\((\d+)\)", x)
                if m:
                    seq.append("w%s" % m.group(1))
        print("%s loads %d stores %d lines %d: %s" % (name, nl, ns, len(b), " ".join(seq)))
