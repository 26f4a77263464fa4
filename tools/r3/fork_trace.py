"""kernel timeline of one replayed training step with and without the backward fork, from rocprofv3 --kernel-trace:
how much of the step two kernels overlap, and what the overlap does to the duration of the kernels involved"""
import csv
import glob
import json
import sys


def step_rows(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    lo, hi = adam[-2] + 1, adam[-1]          # the last full step (a hipGraph replay)
    return rows[lo:hi]


def summarise(rows):
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows])
    depth, last, busy, overlap = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            overlap += t - last
        depth += d
        last = t
    fam = {}
    for s, e, k in rows:
        name = "pw_wgrad" if "pw_wgrad" in k else ("pw_gemm two-tensor (bwd-data)" if "pw_gemm_stream_kernel" in k and ", true," in k
                                                  else ("pw_gemm (forward)" if "pw_gemm" in k else ("dw_march_bwd" if "dw_march_bwd" in k else None)))
        if name:
            a = fam.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += (e - s) / 1e6
    return {"kernels": len(rows), "span_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "two_or_more_kernels_ms": overlap / 1e6,
            "sum_of_durations_ms": sum(e - s for s, e, _ in rows) / 1e6,
            "family_ms": {k: {"launches": v[0], "sum_ms": round(v[1], 3)} for k, v in fam.items()}}


out = {"what": "rocprofv3 --kernel-trace of `bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-split-leg --no-roofline` (B=128, hipGraph), "
               "last replayed step; DL3_FORK=0 (one stream) against DL3_FORK=1 (1x1 weight gradients as a parallel hipGraph branch)",
       "one_stream": summarise(step_rows(sys.argv[1])), "forked": summarise(step_rows(sys.argv[2]))}
print(json.dumps(out, indent=1))
