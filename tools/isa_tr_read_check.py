#!/usr/bin/env python
"""Hazard check for the K-major GEMM's inline-asm LDS transpose reads (csrc/gemm.hip, ds_read_tr16_pair_asm), on the SHIPPED binary.

Those ds_read_b64_tr_b16 are invisible to hipcc's s_waitcnt insertion; the kernel waits lgkmcnt(0) itself before the MFMA block that
consumes the fragments.  Correct only if NOTHING the compiler emitted touches a destination VGPR between the read and that wait -- a
v_mov assembling the 128-bit operand, a spill, a copy of a loop-carried fragment would read registers the LDS has not filled yet.  The
register allocation decides, so it is checked on the disassembly of liblseg_hip.so, every build (tests/test_isa_tr_reads.py, and by hand:
  python tools/isa_tr_read_check.py [path/to/liblseg_hip.so]).

Method: for every kernel instantiated with TAG=2 (K-major) and RELU_IN=0 (the asm form), walk the control-flow graph from each
ds_read_b64_tr_b16 along every path until an `s_waitcnt` whose lgkmcnt field is 0; any instruction on the way that names a destination
register of the read (as source or destination, other than an MFMA-free re-issue of the same read) is a violation.  Paths that reach
s_endpgm without a wait are violations too.  No GPU needed (llvm-objdump)."""
import os, re, shutil, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "lang-seg_amd", "lseg_hip", "liblseg_hip.so")
ASM_KERNEL = re.compile(r"lseg_gemm_kernelINS_\d\w+?ENS0_7TileCfgI[\w]+?EEELb[01]ELb0ELi\d+ELi2EE")   # RELU_IN = 0, TAG = 2


def disassemble(lib):
    """-> text of every gfx950 code object embedded in `lib`."""
    tmp = tempfile.mkdtemp(prefix="isa_tr_")
    try:
        dst = os.path.join(tmp, "lib.so")
        shutil.copy(lib, dst)
        subprocess.run([OBJDUMP, "--offloading", dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                out.append(subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout)
        return "\n".join(out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(dis):
    """-> {symbol: [(addr, mnemonic, operand text, branch target addr | None)]}"""
    funcs, cur, base = {}, None, 0
    for line in dis.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            base, cur = int(m.group(1), 16), []
            funcs[m.group(2)] = cur
            continue
        if cur is None or not line.startswith("\t"):
            continue
        m = re.match(r"^\t(\S+)\s*(.*?)\s*// ([0-9A-F]+): \S+(?: \S+)?(?: <\S+?\+0x([0-9a-f]+)>)?", line)
        if not m:
            continue
        tgt = base + int(m.group(4), 16) if m.group(4) and m.group(1).startswith(("s_cbranch", "s_branch")) else None
        cur.append((int(m.group(3), 16), m.group(1), m.group(2), tgt))
    return funcs


def vregs(text):
    """VGPR numbers named in an operand string: v7, v[4:7]; AGPRs (a…) are a different file."""
    regs = set()
    for m in re.finditer(r"(?<![\w])v\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w\[])v(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def waits_lgkm0(mn, ops):
    if mn != "s_waitcnt":
        return False
    m = re.search(r"lgkmcnt\((\d+)\)", ops)
    return bool(m) and int(m.group(1)) == 0


def check_function(ins):
    """-> list of violations (read addr, offending addr, text)."""
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    bad = []
    for i0, (a0, mn0, ops0, _) in enumerate(ins):
        if mn0 != "ds_read_b64_tr_b16":
            continue
        dst = vregs(ops0.split(",")[0])
        seen, stack = set(), [i0 + 1]
        while stack:
            i = stack.pop()
            while True:
                if i in seen:
                    break
                seen.add(i)
                if i >= len(ins):
                    bad.append((a0, None, "fell off the function without lgkmcnt(0)")); break
                a, mn, ops, tgt = ins[i]
                if waits_lgkm0(mn, ops):
                    break
                if mn == "s_endpgm":
                    bad.append((a0, a, "s_endpgm before lgkmcnt(0)")); break
                if mn == "ds_read_b64_tr_b16":
                    touched = vregs(ops.split(",")[0]) & dst          # a later read INTO the same registers before the wait: lost data
                else:
                    touched = vregs(ops) & dst
                if touched:
                    bad.append((a0, a, f"{mn} {ops}  touches v{sorted(touched)} before lgkmcnt(0)")); break
                if tgt is not None:
                    if tgt in index:
                        stack.append(index[tgt])
                    if mn == "s_branch":
                        break
                i += 1
    return bad


def check_library(lib=DEFAULT_LIB):
    """-> (number of asm kernels, number of asm reads, {symbol: violations})"""
    funcs = functions(disassemble(lib))
    nk = nr = 0
    report = {}
    for name, ins in funcs.items():
        if not ASM_KERNEL.search(name):
            continue
        reads = sum(1 for x in ins if x[1] == "ds_read_b64_tr_b16")
        if not reads:
            continue
        nk += 1
        nr += reads
        v = check_function(ins)
        if v:
            report[name] = v
    return nk, nr, report


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    nk, nr, report = check_library(lib)
    print(f"{lib}: {nk} K-major asm-read kernels, {nr} ds_read_b64_tr_b16")
    for name, v in report.items():
        print(" ", name)
        for a0, a, why in v[:12]:
            print(f"    read @{a0:x}: {'@%x ' % a if a else ''}{why}")
    sys.exit(1 if report else 0)
