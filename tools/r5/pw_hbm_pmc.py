"""Round 5: per-launch PMC of the HBM-bound 1x1-convolution shapes of the benchmarked plan.

    python tools/r5/pw_hbm_pmc.py <microbench log> <pmc dir> [<pmc dir> ...]

The pmc dirs come from `rocprofv3 --kernel-trace --pmc <counters> -- python tools/r5/pw_hbm_bench.py` (one pass per counter
set, tools/collect_round.sh); the log from one such pass.  Every (shape, variant) line of the microbench is REPS + 2
consecutive dispatches of one kernel: the dispatches are grouped in order of Dispatch_Id and matched to the log's lines in
order.  FETCH_SIZE is doubled (gfx950 correction of MI355X_MICROARCH.md, calibrated on wide coalesced reads), WRITE_SIZE as
reported; both in KB per dispatch from rocprofv3."""
import collections
import csv
import glob
import re
import sys


def groups(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("pw_gemm", "pw_fwd_ws", "pw_bwd_fused")):
            continue
        per.setdefault(int(r["Dispatch_Id"]), (k, {}))[1][r["Counter_Name"]] = float(r["Counter_Value"])
    out, last = [], None
    for did, (k, c) in per.items():
        if last is None or last[0] != k:
            last = (k, [])
            out.append(last)
        last[1].append(c)
    return out


def main():
    lines = [l.rstrip() for l in open(sys.argv[1]) if re.match(r"^(fwd|fused) ", l) and "unsupported" not in l]
    merged = None
    for d in sys.argv[2:]:
        g = groups(d)
        if not g:
            print("# %s: no counters collected (counter name not known to this rocprofv3?)" % d)
            continue
        if merged is None:
            merged = [(k, [dict(x) for x in cs]) for k, cs in g]
        else:
            assert len(g) == len(merged), (d, len(g), len(merged))
            for (k, cs), (k2, cs2) in zip(merged, g):
                assert k == k2, (k, k2)
                for a, b in zip(cs, cs2):
                    a.update(b)
    # a kernel that serves two consecutive lines shows up as one group of twice the dispatches: split it evenly
    flat = []
    for k, cs in merged:
        n = max(1, round(len(cs) / PER))
        step = len(cs) // n
        for i in range(n):
            flat.append((k, cs[i * step:(i + 1) * step]))
    print("# %d microbench lines, %d dispatch groups" % (len(lines), len(flat)))
    for i, (k, cs) in enumerate(flat):
        line = lines[i] if i < len(lines) else "?"
        m = re.search(r"([0-9.]+) ms +([0-9]+) GB/s", line)
        shape = line.split("  ")[0]
        mean = lambda c: sum(x.get(c, 0.0) for x in cs[2:] or cs) / max(1, len(cs[2:] or cs))
        fetch, write = 2.0 * mean("FETCH_SIZE") * 1024.0, mean("WRITE_SIZE") * 1024.0
        alg = float(m.group(2)) * float(m.group(1)) * 1e6 if m else 0.0
        kn = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        extra = " ".join("%s=%.3g" % (c, mean(c)) for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA_WRREQ_sum", "TCC_EA_WRREQ_64B_sum", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES",
                                                          "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if any(c in x for x in cs))
        print("%-62s | %-34s n=%d | fetch %7.1f MB write %7.1f MB | algorithmic %7.1f MB | traffic/alg %.2f | %s" % (
            shape[:62], kn[:34], len(cs), fetch / 1e6, write / 1e6, alg / 1e6, (fetch + write) / alg if alg else 0.0, extra))


PER = 5  # REPS=3 + 2 warm-up launches per line (tools/collect_round.sh sets REPS=3)
if __name__ == "__main__":
    main()
