"""scan kernel ISA for the store-drain pattern of round 5: inside a loop that contains global stores, an
`s_waitcnt vmcnt(0)` that sits in front of the loop body's first global load (the wave waits for its own stores before it
even asks for its next rows).  usage: python tools/r5/asm_drain_scan.py file.s ..."""
import re
import subprocess
import sys

for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith("_Z") and lines[i].rstrip().endswith(":") or (lines[i].startswith("_Z") and ":" in lines[i]):
            name = lines[i].split(":")[0]
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            body = lines[i:j]
            dn = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
            dn = dn.replace("(anonymous namespace)::", "")[:70]
            # innermost loops: from a "Loop Header" label to the backward branch to it
            for k, l in enumerate(body):
                m = re.match(r"^(\.LBB\d+_\d+):", l)
                if not m:
                    continue
                hdr = l
                q = k + 1
                while q < len(body) and body[q].lstrip().startswith(";"):
                    hdr += body[q]
                    q += 1
                if "Loop Header" not in hdr:
                    continue
                lab = m.group(1)
                end = None
                for q in range(k + 1, len(body)):
                    if re.search(r"s_cbranch_\w+ " + re.escape(lab) + r"\b", body[q]) or re.search(r"s_branch " + re.escape(lab) + r"\b", body[q]):
                        end = q
                if end is None:
                    continue
                loop = body[k:end + 1]
                nst = sum("global_store" in x for x in loop)
                nld = sum("global_load" in x for x in loop)
                if nst == 0 or nld == 0:
                    continue
                first_ld = next(q for q, x in enumerate(loop) if "global_load" in x)
                pre = [x for x in loop[:first_ld] if "s_waitcnt" in x and "vmcnt(0)" in x]
                tot0 = sum(("s_waitcnt" in x and "vmcnt(0)" in x) for x in loop)
                flag = "DRAIN-BEFORE-LOADS" if pre else ""
                print("%-72s loop %-10s %4d lines  loads %3d stores %3d  vmcnt(0) %2d %s" % (dn, lab, len(loop), nld, nst, tot0, flag))
            i = j
        else:
            i += 1
