"""outline of one kernel's ISA: labels, branches, barriers, scratch traffic, MFMA / global / LDS instruction counts per
basic block.  usage: python tools/r5/asm_outline.py file.s <substring of the mangled kernel name>"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start + 1, len(s)) if s[i].startswith(".Lfunc_end"))
blk, cnt = "entry", {}
def flush():
    if cnt:
        print("  %-10s %s" % (blk, " ".join("%s=%d" % kv for kv in sorted(cnt.items()))))
for l in s[start + 1:end]:
    t = l.strip()
    if t.startswith(".LBB"):
        flush()
        blk, cnt = t.split(":")[0], {}
        continue
    op = t.split(" ")[0]
    for k, pat in (("mfma", "v_mfma"), ("gload", "global_load"), ("gstore", "global_store"), ("dsr", "ds_read"), ("dsw", "ds_write"),
                   ("scr_st", "scratch_store"), ("scr_ld", "scratch_load"), ("bar", "s_barrier"), ("br", "s_cbranch"), ("valu", "v_")):
        if op.startswith(pat):
            cnt[k] = cnt.get(k, 0) + 1
            break
flush()
