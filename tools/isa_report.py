#!/usr/bin/env python
"""What hipcc made of the GEMM kernel family, per instantiation, WITHOUT a GPU: registers, scratch bytes (spills), and every s_waitcnt
vmcnt(N) inside the K-step loop and behind it (epilogue).  Round 4's three largest kernel gains came from this listing, not from a
profile: a compiler-inserted `s_waitcnt vmcnt(0)` is a full drain of the direct-to-LDS ring, and they appear silently --
  * behind every scratch reload (a spilling epilogue puts them into the K-loop's tile-advance paths: EPI_RES32, -16 us per launch),
  * at the top of every conditional store block that uses loaded data (EPI_QKV16: four store round trips per tile, +23 us per launch),
  * in front of ds_read_b64_tr_b16 builtins while direct-to-LDS loads are pending (K-major weight-gradient GEMM: twice per K-step).
  python tools/isa_report.py [substring ...]      e.g.  F16 256x256 conv0      (all substrings must match the instance key)
  python tools/isa_report.py --file attention.hip --flags=-fno-slp-vectorize
A kernel line reads:  <dtype> <tile> <waves> conv<0|1> relu<0|1> epi<EPI> tag<TAG>  vgpr N  scratch BYTES | waits in loop [...] after [...]
In the K-loop only the counted waits of the loop itself (vmcnt(A_SPW) per barrier) should show; any 0 there is a drain."""
import argparse, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--file", default="gemm.hip"); ap.add_argument("--flags", default=""); ap.add_argument("--asm", default="", help="reuse / keep this .s file")
ap.add_argument("match", nargs="*")
a = ap.parse_args()
src = os.path.join(ROOT, "lang-seg_amd", "csrc", a.file)
out = a.asm or os.path.join(tempfile.gettempdir(), os.path.basename(a.file) + ".s")
if not (a.asm and os.path.exists(out)):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-unused-variable",
           "--cuda-device-only", "-S", src, "-o", out] + a.flags.split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    g = re.search(r"lseg_gemm_kernelINS_\d(\w+?)ENS0_7TileCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+EEELb(\d)ELb(\d)ELi(\d+)ELi(\d+)", name)
    key = (f"{g.group(1)} {g.group(2)}x{g.group(3)} {g.group(4)}x{g.group(5)} conv{g.group(6)} relu{g.group(7)} epi{g.group(8)} tag{g.group(9)}"
           if g else re.sub(r"^_ZN4lseg12_GLOBAL__N_1\d+", "", name)[:70])
    if a.match and not all(w in key for w in a.match):
        continue
    val = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
    i = s.index("\n" + name + ":"); j = s.index(".Lfunc_end", i)
    L = s[i:j].split("\n")
    deep = [k for k, l in enumerate(L) if "Depth=2" in l] or [k for k, l in enumerate(L) if "Depth=1" in l]
    lo, hi = (deep[0], deep[-1]) if deep else (0, 0)
    waits = [(k, re.search(r"vmcnt\((\d+)\)", l).group(1)) for k, l in enumerate(L) if "s_waitcnt" in l and "vmcnt" in l]
    print(key, "vgpr", val("next_free_vgpr"), "scratch", val("private_segment_fixed_size"), "| waits in loop",
          [w for k, w in waits if lo <= k <= hi], "after", [w for k, w in waits if k > hi])
