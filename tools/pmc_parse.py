"""Parses rocprofv3 --pmc counter_collection CSVs for the dw_march kernels -> JSON summary (bytes per launch)."""
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dw_march" not in k:
                continue
            name = "dw_march_fwd" if ("dw_march_fwd" in k or "dw_march2_fwd" in k) else "dw_march_bwd"
            acc.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (name, ctr), vals in acc.items():
            vals = vals[3:] if len(vals) > 6 else vals  # drop warm-up launches
            out.setdefault(name, {})[ctr] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
print(json.dumps(out, indent=1))
