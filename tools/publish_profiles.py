"""copies the summaries tools/collect_round.sh left under gpurun_out/<OUTDIR>/ into profiles/ (rNN_ names) and writes the
README next to every kernel-stats table (per-step family times from the trace next to the in-situ figures of the same run)

    python tools/publish_profiles.py <round number> [OUTDIR, default final]"""
import csv
import json
import re
import shutil
import sys

RND = int(sys.argv[1])
R = "r%02d" % RND
S = "gpurun_out/%s/" % (sys.argv[2] if len(sys.argv) > 2 else "final")
P = "profiles/"
COPY = {"cfg2_b128_gemm_pmc.json": R + "_gemm_b128_pmc.json", "cfg2_b128_dw_dilated_pmc.json": R + "_dw_dilated_b128_pmc.json",
        "cfg4_b16_gemm_pmc.json": R + "_gemm_b16_pmc.json", "cfg4_b16_dw_dilated_pmc.json": R + "_dw_dilated_b16_pmc.json",
        "cfg2_b128_kernel_stats.csv": R + "_bench_default_b128_kernel_stats.csv",
        "cfg3_subpixel_b128_kernel_stats.csv": R + "_bench_cfg3_subpixel_b128_kernel_stats.csv",
        "cfg4_b16_kernel_stats.csv": R + "_bench_cfg4_xception_os8_b16_kernel_stats.csv",
        "cfg2_split_b128_kernel_stats.csv": R + "_bench_split_b128_kernel_stats.csv",
        "plan_cfg2_b128.json": R + "_insitu_plan_b128.json", "plan_cfg4_b16.json": R + "_insitu_plan_cfg4_b16.json",
        "plan_cfg3_subpixel_b128.json": R + "_insitu_plan_cfg3_subpixel_b128.json",
        "cfg2_b16_kernel_stats.csv": R + "_bench_cfg2_b16_kernel_stats.csv", "plan_cfg2_b16.json": R + "_insitu_plan_b16.json",
        "cfg2_b2_kernel_stats.csv": R + "_bench_cfg2_b2_kernel_stats.csv", "plan_cfg2_b2.json": R + "_insitu_plan_b2.json"}
for a, b in COPY.items():
    shutil.copy(S + a, P + b)
with open(P + R + "_bench_lines.jsonl", "w") as f:
    f.write(open(S + "bench_default.json").read().strip().splitlines()[-1] + "\n")
    f.write(open(S + "bench_lines.jsonl").read())
hdr = open(P + "r02_bench_b128_sq_pmc.txt").read().split("\n")[:4]
with open(P + R + "_bench_b128_sq_pmc.txt", "w") as f:
    f.write("\n".join(h.replace("round 2 (final tree)", "round %d (final tree)" % RND).replace("collect_sq_pmc.sh 128", "collect_round.sh") for h in hdr) + "\n")
    f.write(open(S + "cfg2_b128_sq_pmc.txt").read())


def fam(t):
    rows = list(csv.DictReader(open(S + t + "_kernel_stats.csv")))
    steps = [int(x["Calls"]) for x in rows if "adam_kernel" in x["Name"]][0]
    tot = sum(float(x["TotalDurationNs"]) for x in rows)
    g = sum(float(x["TotalDurationNs"]) for x in rows if any(t in x["Name"] for t in ("pw_gemm", "pw_ksplit32", "pw_fwd_ws", "pw_ws2", "pw_narrowk", "pw_rows_f64", "pw_wgrad", "pw_bwd_fused")))
    d = sum(float(x["TotalDurationNs"]) for x in rows if "dw_march" in x["Name"])
    return steps, tot, g, d


def insitu(plan):
    rows = json.load(open(S + plan))["rows"]
    g = sum(r["ms"] for r in rows if r["family"] == "gemm")
    d = sum(r["ms"] for r in rows if r["op"].startswith("dl3_dwconv3x3") and " s1 " in r["shape"] + " ")
    return g, d, sum(r["ms"] for r in rows)


for tag, plan, cmd in (("cfg2_b128", "plan_cfg2_b128.json", "python bench.py --no-cpu-baseline --no-split-leg --steps 10 --warmup 3 --plan-json ..."),
                       ("cfg3_subpixel_b128", "plan_cfg3_subpixel_b128.json", "python bench.py --head subpixel --no-cpu-baseline --no-split-leg --steps 10 --warmup 3 --plan-json ..."),
                       ("cfg4_b16", "plan_cfg4_b16.json", "python bench.py --backbone xception --os 8 --batch 16 --no-cpu-baseline --no-split-leg --steps 6 --warmup 3 --plan-json ..."),
                       ("cfg2_b16", "plan_cfg2_b16.json", "python bench.py --batch 16 --no-cpu-baseline --no-split-leg --steps 10 --warmup 3 --plan-json ..."),
                       ("cfg2_b2", "plan_cfg2_b2.json", "python bench.py --batch 2 --no-cpu-baseline --no-split-leg --steps 20 --warmup 3 --plan-json ..."),
                       ("cfg2_split_b128", "plan_cfg2_split_b128.json", "DL3_GEMM_MATH=split python bench.py --no-cpu-baseline --no-split-leg --steps 10 --warmup 3 --plan-json ...")):
    steps, tot, g, d = fam(tag)
    ig, idw, it = insitu(plan)
    timed = re.findall(r"timed \d+ steps: ([0-9.]+) ms/step", open(S + "prof_%s.log" % tag).read())[-1]
    n = steps + 3
    name = {"cfg2_b128": R + "_bench_default_b128", "cfg3_subpixel_b128": R + "_bench_cfg3_subpixel_b128",
            "cfg4_b16": R + "_bench_cfg4_xception_os8_b16", "cfg2_split_b128": R + "_bench_split_b128",
            "cfg2_b16": R + "_bench_cfg2_b16", "cfg2_b2": R + "_bench_cfg2_b2"}[tag]
    txt = ("# rocprofv3 --kernel-trace --stats of `%s` (tools/collect_round.sh), final tree of round %d\n"
           "# %d steps in the trace: %d with adam_kernel (warm-ups + timed hipGraph replays) + 3 eager in-situ passes; per-step device time by\n"
           "# kernel-name family (TotalDurationNs / %d) next to bench.py's in-situ HIP-event figures of the SAME run:\n"
           "#   1x1-conv GEMM kernels (pw_gemm_* + pw_fwd_ws_* + pw_wgrad_* + pw_bwd_fused*): %.2f ms/step | in situ (the family's ops + the pass's one slab-fold launch): %.2f ms/step\n"
           "#   depthwise march kernels (dw_march*):            %.2f ms/step | in situ (stride-1 depthwise ops): %.2f ms/step\n"
           "#   all kernels: %.2f ms/step | in situ all launches %.2f ms | timed region of this run: %s ms/step\n") % (
               cmd, RND, n, steps, n, g / 1e6 / n, ig, d / 1e6 / n, idw, tot / 1e6 / n, it, timed)
    open(P + "%s_kernel_stats.README.txt" % name, "w").write(txt)
    print(txt)
