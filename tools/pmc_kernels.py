"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel: mean of each counter over dispatches."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    n = len(next(iter(d.values())))
    rows.append((sum(d.get("SQ_WAVE_CYCLES", d.get("GRBM_GUI_ACTIVE", [0]))), k, n, {c: sum(v) / len(v) for c, v in d.items()}))
for tot, k, n, m in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%-44s n=%4d " % (k[:44], n) + " ".join("%s=%.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(m.items())))
