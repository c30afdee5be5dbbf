#!/usr/bin/env python
"""profiles/<round>_traffic.json (argv[3], default profiles/r02_traffic.json) from the PMC summary of tools/collect_profiles.sh (gpurun_out/profiles/summary.json).
HBM-side traffic of the profiled MLP fc1 GEMM (the TAG=1 kernel symbol), per launch:
  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM/rocprofv3 section)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "profiles", "summary.json")
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 36
pmc = json.load(open(src))["pmc"]
name = [k for k in pmc if "lseg_gemm_kernel" in k and k.rstrip().endswith(", 1>(lseg::GemmArgs)")]
# the bench's self-check adds one single-image forward (a smaller tile configuration of the same tagged kernel):
# the timed batch is the instance with the larger per-launch MFMA count
name.sort(key=lambda k: -pmc[k].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0))
c = pmc[name[0]]
M, N, K = batch * 901, 4096, 1024
out = {"mlp_fc1_gemm": {
    "kernel": name[0], "batch": batch,
    "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "traffic_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); FETCH counts L2->fabric requests incl. Infinity-Cache hits",
    "algorithmic_bytes_per_launch": 2 * (M * K + N * K + M * N),
    **{k: c[k] for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_LDS_BANK_CONFLICT") if k in c}}}
# the correlation kernel (labels as GEMM rows, 160x128 tiles) and the two attention kernels, when profiled
def add(key, pred, alg_bytes=None):
    names = [k for k in pmc if pred(k)]
    if not names:
        return
    names.sort(key=lambda k: -(pmc[k].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) + pmc[k].get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0)))
    c = pmc[names[0]]
    e = {"kernel": names[0], **{k: c[k] for k in c}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["traffic_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if alg_bytes:
        e["algorithmic_bytes_per_launch"] = alg_bytes
    out[key] = e
P = batch * 57600
add("correlation_gemm", lambda k: "lseg_gemm_kernel" in k and "TileCfg<160, 128" in k, 2 * P * 512 + 150 * 512 * 2 + 4 * P * 150)
add("attention", lambda k: "lseg_attention_kernel<lseg::BF16>" in k)
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r02_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
