"""HBM traffic of the dilated-depthwise launches over whole training steps, from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE; separate runs, MI355X_MICROARCH.md "HBM" section) of `bench.py --no-graph --no-roofline --no-cpu-baseline`.

usage: python tools/pmc_family.py <fetch_dir> <write_dir> <plan.json> > profiles/r02_dw_dilated_b<B>_pmc.json

The counters are per dispatch (KB).  Dispatches are split into steps at the adam_kernel launches (one per step).  The
march kernels (dw_march_fwd / dw_march2_fwd / dw_march_bwd) serve every stride-1 depthwise layer, rate 1 included, so
the dilated launches are picked by ORDER: the i-th dw_march* dispatch of a step is the i-th stride-1 depthwise row of
the plan bench.py wrote with --plan-json (same engine, same launch sequence), whose family field says whether it is
dilated; the algorithmic bytes in the output are summed over exactly those rows.
gfx950 correction (same section of the guide): FETCH_SIZE counts a wide coalesced read at half its bytes -> doubled;
WRITE_SIZE is taken as reported.
"""
import csv
import glob
import json
import sys


def per_step(d, march_rows):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    steps, cur = [], []
    for _, k, v in rows:
        if "dw_march" in k:
            cur.append(v)
        elif "adam_kernel" in k:
            steps.append(cur)
            cur = []
    good = [s for s in steps if len(s) == len(march_rows)]
    assert good, "no step with %d dw_march dispatches (saw %s)" % (len(march_rows), [len(s) for s in steps])
    tot = [sum(v for v, r in zip(s, march_rows) if r["family"] == "dw_dilated") for s in good]
    return sum(tot) / len(tot), len(good)


def gemm_family(fetch_dir, write_dir, plan):
    """1x1-conv GEMM family by kernel name (pw_gemm_* + pw_fwd_ws_* + pw_wgrad_* + pw_bwd_fused*): per-step sums, steps split at
    adam_kernel"""
    def load(d):
        tot, steps = 0.0, 0
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                steps += "adam_kernel" in k
                if any(t in k for t in ("pw_gemm", "pw_ksplit32", "pw_wgrad", "pw_bwd_fused", "pw_fwd_ws", "pw_ws2", "pw_narrowk", "pw_rows_f64")):
                    tot += float(r["Counter_Value"])
        return tot / max(steps, 1), steps
    f, fs = load(fetch_dir)
    w, ws = load(write_dir)
    alg = sum(r["bytes"] for r in plan["rows"] if r["family"] == "gemm")
    fetch, write = 2.0 * f * 1024.0, w * 1024.0
    print(json.dumps({
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --no-graph "
                "--no-roofline --no-cpu-baseline; per-step sum over the pw_gemm_* / pw_fwd_ws_* / pw_wgrad_* / pw_bwd_fused* dispatches; FETCH_SIZE "
                "doubled (gfx950 correction of MI355X_MICROARCH.md; calibrated on wide coalesced reads — the GEMM A "
                "operand is read in 16-byte pieces per lane, so treat the absolute as approximate), WRITE_SIZE as reported. "
                "Algorithmic bytes count every operand once per launch; column tiles of a row tile re-read the activation "
                "tile (L2 hits are not HBM traffic, misses are).",
        "family": "gemm", "batch": plan["batch"], "backbone": plan["backbone"], "steps_used": [fs, ws],
        "fetch_bytes_per_step": fetch, "write_bytes_per_step": write, "traffic_bytes_per_step": fetch + write,
        "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": (fetch + write) / alg}, indent=1))


def main():
    if len(sys.argv) > 4 and sys.argv[4] == "gemm":
        return gemm_family(sys.argv[1], sys.argv[2], json.load(open(sys.argv[3])))
    fetch_dir, write_dir, plan = sys.argv[1], sys.argv[2], json.load(open(sys.argv[3]))
    march_rows = [r for r in plan["rows"] if r["op"].startswith("dl3_dwconv3x3") and " s1 " in r["shape"] + " "]
    dil = [r for r in march_rows if r["family"] == "dw_dilated"]
    f, fs = per_step(fetch_dir, march_rows)
    w, ws = per_step(write_dir, march_rows)
    fetch, write = 2.0 * f * 1024.0, w * 1024.0
    alg = sum(r["bytes"] for r in dil)
    print(json.dumps({
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --no-graph "
                "--no-roofline --no-cpu-baseline; per-step sum over the dilated depthwise dispatches (picked by launch "
                "order against the --plan-json rows); FETCH_SIZE doubled (gfx950 correction of MI355X_MICROARCH.md), "
                "WRITE_SIZE as reported",
        "family": "dw_dilated", "batch": plan["batch"], "backbone": plan["backbone"], "steps_used": [fs, ws],
        "launches_per_step": len(dil), "fetch_bytes_per_step": fetch, "write_bytes_per_step": write,
        "traffic_bytes_per_step": fetch + write, "algorithmic_bytes_per_step": alg,
        "traffic_over_algorithmic": (fetch + write) / alg}, indent=1))


if __name__ == "__main__":
    main()
