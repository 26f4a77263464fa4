"""Generates KNOBS.md: every DL3_* environment variable the package, the library, bench.py and __graft_entry__.py read — name,
default, where it is read, and what kind of switch it is.  The scan finds the variables; the descriptions live here, and
tests/test_host.py fails when the two part (a knob without a description, a description without a knob).

  python tools/knobs.py          # rewrites KNOBS.md
  python tools/knobs.py --check  # exit 1 if KNOBS.md is stale"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "keras-segmentation-deeplab-v3.1_amd")
PAT = re.compile(r"(?:getenv|env_int|fused_env)\(\"(DL3_[A-Z0-9_]+)\"|environ(?:\.get|\.pop|\.setdefault)?[\(\[]\"(DL3_[A-Z0-9_]+)\"")

# name -> (default, kind, meaning).  kind: semantic = changes what is computed / where files are found; data plane = multi-GPU
# transport; resource = memory policy; test aid = exists for a parity / robustness test; A/B = keeps a measured alternative
# reachable for same-call comparisons (never needed for the default, benchmarked behaviour)
KNOBS = {
    "DL3_WEIGHTS_DIR": ("(unset)", "semantic", "directory holding deeplabv3_{mobilenetv2,xception}_tf_dim_ordering_tf_kernels.h5 for weights='pascal_voc' (deeplabv3p.py:458-464; no download here)"),
    "DL3_GEMM_MATH": ("f32", "semantic", "'split': 1x1-convolution GEMMs as 3 x bf16 pieces on the bf16 matrix pipe (opt-in, never the headline); dl3_set_gemm_math overrides"),
    "DL3_DIST_BACKEND": ("rccl on a GPU box", "data plane", "'gloo': gradients staged through host memory (the CPU tests and the two-processes-on-one-GPU tests)"),
    "DL3_DIST_STRICT": ("0 (Model.distribute(): strict=True)", "data plane", "1: a failed RCCL bring-up raises instead of falling back to gloo; 0 overrides Model.distribute()'s strict default"),
    "DL3_MAX_ENGINES": ("4", "resource", "engines (one activation arena each) a Model keeps before the least recently used is dropped"),
    "DL3_LIBPATH": ("in-tree libdl3.so", "A/B", "another build of the library (build_variants/, tools/ab.sh)"),
    "DL3_POISON_SCRATCH": ("0", "test aid", "1: every scratch allocation pre-filled with NaN (test_poisoned_scratch_changes_nothing, test_benchmarked_plan_*)"),
    "DL3_BATCH_FOLDS": ("1", "test aid", "0: one fold launch per weight gradient instead of the batched one (tested bit-identical)"),
    "DL3_DY_MAT": ("1", "test aid", "0: two-tensor gradient operand in every bwd-data GEMM (no dY written by the weight-gradient launch); the backward-fork test needs it"),
    "DL3_FORK": ("0", "A/B", "1 / 2: weight gradients on a second captured stream / paired with depthwise backward launches (round 3: -6 % / +0.5 %; tested bit-identical)"),
    "DL3_FUSED_BWD": ("1", "test aid", "0: separate weight-gradient and bwd-data launches for the HBM-bound early layers"),
    "DL3_FUSED_ROWS": ("32768", "test aid", "minimum pixel rows of a layer for the both-gradient kernel (tests lower it to run the kernel inside small plans)"),
    "DL3_FUSED_V": ("2", "A/B", "1: the round-4 both-gradient kernel (op tests keep it alive)"),
    "DL3_FUSE_SHUFFLE": ("1", "test aid", "0: Subpixel head with materialised phase shift + plain loss (the fused loss is tested against it)"),
    "DL3_PRUNE_BWD": ("1", "test aid", "0: lower the whole backward pass even below the first trainable parameter"),
    "DL3_GEMM_CFG": ("(cost model)", "test aid", "force one of the 7 tile configurations of the 1x1 GEMMs (test_pwconv_every_tile_configuration, tools/gemm_tune.py)"),
    "DL3_WGRAD_CFG": ("(cost model)", "test aid", "force one of the 10 tile configurations of the weight-gradient kernel"),
    "DL3_GEMM_PRE": ("1", "test aid", "0: no prefetching 128x96 bwd-data tile (op tests run both)"),
    "DL3_DW_PPB": ("(heuristic)", "test aid", "row phases per workgroup of the dilated depthwise kernels (forced multi-phase decompositions in the op tests)"),
    "DL3_XENT_PREF": ("1", "test aid", "0: the loss row kernel requests each row's operands when the row starts (tested bit-identical)"),
    "DL3_GEMM_PY": ("512 / 2048 / 4096 by shape", "A/B", "target number of workgroups of a 1x1 GEMM launch (round 6: profiles/r06_ab_calls.txt call 6)"),
    "DL3_WS2": ("1", "A/B", "0: no weight-stationary kernel for the MFMA-bound short reductions (round 6, calls 12 / 14)"),
    "DL3_NARROW": ("1", "A/B", "0: the logits layer (N = classes off a 256-wide input) on the tiled kernels (round 6, call 24: forward / bwd-data / weight gradient 0.166 / 0.222 / 0.339 -> 0.124 / 0.112 / 0.139 ms at 524 288 rows)"),
    "DL3_COLSPLIT": ("1", "A/B", "0: a 736-wide output (Xception's 728 channels, 23 column blocks) in one launch of six 128-wide tiles instead of 256 + 480 columns in two (round 6, calls 26 / 27: cfg4 87.3 -> 88.6 img/s)"),
    "DL3_WGRAD_ROW": ("1", "A/B", "0: the expand convolutions' weight gradient on the tiled kernel (round 6, call 18)"),
}


def scan():
    found = {}
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for d, _, fs in os.walk(PKG):
        files += [os.path.join(d, f) for f in fs if f.endswith((".py", ".hip", ".cpp", ".h"))]
    for f in sorted(files):
        for i, line in enumerate(open(f, errors="ignore"), 1):
            for m in PAT.finditer(line):
                site = os.path.relpath(f, ROOT).replace("keras-segmentation-deeplab-v3.1_amd/", "")
                sites = found.setdefault(m.group(1) or m.group(2), [])
                if site not in sites:   # (files, not lines: the table must not go stale with every edit)
                    sites.append(site)
    return found


def render(found):
    out = ["# DL3_* environment variables", "",
           "Generated by `python tools/knobs.py` (tests/test_host.py keeps it current).  The defaults are what is benchmarked and",
           "tested; no variable is needed for the reference-compatible behaviour.  Round 6 removed 27 tuning aids whose A/B is",
           "settled (HISTORY.md has their measurements): the values they defaulted to are constants now.", "",
           "| variable | default | kind | read at | meaning |", "|---|---|---|---|---|"]
    order = {"semantic": 0, "data plane": 1, "resource": 2, "test aid": 3, "A/B": 4}
    for name in sorted(found, key=lambda n: (order[KNOBS[n][1]], n)):
        d, kind, what = KNOBS[name]
        sites = found[name]
        out.append("| `%s` | %s | %s | %s | %s |" % (name, d, kind, ", ".join("`%s`" % s for s in sites[:3]) + (" …" if len(sites) > 3 else ""), what))
    out += ["", "Build-time macros of `csrc/` (probe builds under `build_variants/`, never the shipped library): `DL3_PHASE_TIMING`,",
            "`DL3_STREAM_KT_FWD`, `DL3_STREAM_KT_BWD1`, `DL3_STREAM_PD_SMALL`, `DL3_WGRAD_MS`, `DL3_WS2_NW`, `DL3_WS2_DIRECT_EPILOGUE`,",
            "`DL3_DBG_NOSTORE`, `DL3_DBG_NOSTAT`, `DL3_EPI_NOFENCE`, `DL3_ACT_MINMAX`.", ""]
    return "\n".join(out)


if __name__ == "__main__":
    found = scan()
    missing, stale = sorted(set(found) - set(KNOBS)), sorted(set(KNOBS) - set(found))
    if missing or stale:
        print("knobs without a description: %s; descriptions without a knob: %s" % (missing, stale))
        sys.exit(1)
    text = render(found)
    path = os.path.join(ROOT, "KNOBS.md")
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print("KNOBS.md: %d variables" % len(found))
