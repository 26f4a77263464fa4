"""profiles/<tag>_dw_r4_b<B>_pmc.json from the raw PMC means (tools/pmc_parse.py output): HBM bytes per launch of the
dilated depthwise kernels, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950.
usage: python tools/pmc_summary.py gpurun_out/prof_r01_pmc_raw.json 32 > profiles/r01_dw_r4_b32_pmc.json"""
import json
import sys

raw = json.load(open(sys.argv[1]))
B = int(sys.argv[2])
elems = B * 64 * 64 * 960
alg = {"dw_march_fwd": 4.0 * (2 * elems + 9 * 960), "dw_march_bwd": 4.0 * (4 * elems + 2 * 9 * 960)}
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/dw_only.py %d "
               "(tools/collect_profiles.sh); counters are KB per dispatch, mean over the launches after warm-up. gfx950 "
               "correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of a wide coalesced "
               "(16 B/lane) streaming read -> doubled; WRITE_SIZE taken as is." % B,
       "batch": B, "shape": "%dx64x64x960 rate 4" % B}
for k, a in alg.items():
    f = raw[k]["FETCH_SIZE"]["mean"] * 1024.0
    w = raw[k]["WRITE_SIZE"]["mean"] * 1024.0
    out[k] = {"fetch_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes": w, "traffic_bytes": 2 * f + w,
              "algorithmic_bytes": a, "traffic_over_algorithmic": (2 * f + w) / a, "launches": raw[k]["FETCH_SIZE"]["launches"]}
print(json.dumps(out, indent=1))
