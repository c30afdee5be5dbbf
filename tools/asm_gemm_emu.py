#!/usr/bin/env python
"""Lane-accurate emulator + hazard checker for the generated GEMM body (lang-seg_amd/csrc/gemm_asm_gen.py), CPU only.

Runs the exact instruction text the kernel is built from on a small problem -- 4 waves per workgroup, 64 lanes, SGPR / VGPR / AGPR files,
LDS, a flat global memory -- and compares C with numpy.  Memory operations complete instantly in the VALUE model; the ORDER model beside
it replays what the hardware guarantees and flags what it does not:
  * every wave keeps its in-order VMEM queue; s_waitcnt vmcnt(N) retires all but the youngest N;
  * a direct-to-LDS piece may be read by its own wave once retired, by other waves once retired AND a barrier has passed since;
  * a piece may be overwritten only when every read of it (any wave) happened before the writer's last barrier and has been waited for;
  * a register written by a global_load may be read (MFMA, store) only once retired;
  * every barrier is reached with lgkmcnt = 0 (the kernel's own rule).
tests/test_asm_gemm_emu.py runs it in the CPU suite."""
import os, re, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd", "csrc"))
import gemm_asm_gen as G  # noqa: E402

M32 = 0xffffffff


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(f):
    u = f.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7fff + ((u >> 16) & 1)
    return ((u >> 16) & 0xffff).astype(np.uint16)


class Wave:
    def __init__(self, wid, wg):
        self.wid, self.wg = wid, wg
        self.s = [0] * 106
        self.scc = 0
        self.vcc = np.zeros(64, bool)
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.epoch = 0
        self.vmq = []            # in-order queue of outstanding VMEM ops: dicts
        self.lgkm = []           # outstanding ds_reads: (piece ids)
        self.pending_reg = {}    # ("a"|"v", idx) -> op still in flight


class Emu:
    def __init__(self, lines, dt, mem, kernarg_addr, grid):
        self.dt = dt
        self.mem = mem
        self.kernarg = kernarg_addr
        self.grid = grid
        self.prog = []
        self.labels = {}
        for l in lines:
            l = l.split(";")[0].strip().replace("%=", "")
            if not l:
                continue
            if l.endswith(":"):
                self.labels[l[:-1]] = len(self.prog)
                continue
            self.prog.append(l)
        self.errors = []
        self.stats = {"mfma": 0, "instr": 0}

    # ---- operand helpers -------------------------------------------------------------------------------------------------------------
    def sval(self, w, tok):
        tok = tok.strip()
        if tok == "%1":
            return w.wg
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return w.s[int(m.group(1))]
        if tok == "m0":
            return w.m0
        if tok == "vcc":
            raise ValueError
        return int(tok, 0) & M32

    def spair(self, w, tok):
        tok = tok.strip()
        if tok == "%0":
            return self.kernarg
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        lo = int(m.group(1))
        return w.s[lo] | (w.s[lo + 1] << 32)

    def sset(self, w, tok, val):
        tok = tok.strip()
        if tok == "m0":
            w.m0 = val & M32
            return
        w.s[int(re.fullmatch(r"s(\d+)", tok).group(1))] = val & M32

    def vsrc(self, w, tok):
        """32-bit per-lane source: vN, sN, literal, %2"""
        tok = tok.strip()
        if tok == "%2":
            return (np.arange(64, dtype=np.uint32) + np.uint32(64 * w.wid))
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            self.check_reg(w, "v", int(m.group(1)), 1)
            return w.v[int(m.group(1))].copy()
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return np.full(64, w.s[int(m.group(1))], np.uint32)
        return np.full(64, int(tok, 0) & M32, np.uint32)

    @staticmethod
    def rng(tok):
        tok = tok.strip()
        m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
        if m:
            return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        m = re.fullmatch(r"([va])(\d+)", tok)
        return m.group(1), int(m.group(2)), 1

    def regfile(self, w, kind):
        return w.v if kind == "v" else w.a

    def check_reg(self, w, kind, lo, n):
        for i in range(lo, lo + n):
            if (kind, i) in w.pending_reg:
                self.errors.append(f"wave {w.wid} pc {w.pc} `{self.prog[w.pc]}`: reads {kind}{i} while its load is still in flight")
                del w.pending_reg[(kind, i)]

    # ---- memory ------------------------------------------------------------------------------------------------------------------------
    def gload(self, addr, nbytes):
        out = np.zeros((64, nbytes), np.uint8)
        for l in range(64):
            a = int(addr[l])
            assert 0 <= a and a + nbytes <= self.mem.size, f"global access out of range: {a:#x}"
            out[l] = self.mem[a:a + nbytes]
        return out

    # ---- one instruction -----------------------------------------------------------------------------------------------------------
    def step(self, w, lds, lds_state):
        ins = self.prog[w.pc]
        self.stats["instr"] += 1
        op, _, rest = ins.partition(" ")
        args = [x.strip() for x in re.split(r",(?![^\[]*\])", rest)] if rest else []
        nxt = w.pc + 1
        S = lambda t: self.sval(w, t)
        if op == "s_nop":
            pass
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                while len(w.vmq) > n:
                    o = w.vmq.pop(0)
                    if o["kind"] == "dma":
                        st = lds_state[o["piece"]]
                        if st.get("seq") == o["seq"]:
                            st["landed"] = True
                            st["landed_epoch"] = w.epoch
                    elif o["kind"] == "ld":
                        for r in o["regs"]:
                            if w.pending_reg.get(r) is o:
                                del w.pending_reg[r]
            if "lgkmcnt(0)" in rest:
                w.lgkm = []
        elif op == "s_barrier":
            if w.lgkm:
                self.errors.append(f"wave {w.wid} reaches a barrier with fragment reads in flight (pc {w.pc})")
            w.at_barrier = True
        elif op in ("s_load_dwordx8", "s_load_dwordx16"):
            n = 8 if op.endswith("x8") else 16
            lo = int(re.match(r"s\[(\d+):", args[0]).group(1))
            base = self.spair(w, args[1]) + int(args[2], 0)
            words = self.mem[base:base + 4 * n].view(np.uint32)
            for i in range(n):
                w.s[lo + i] = int(words[i])
        elif op == "s_mov_b32":
            self.sset(w, args[0], S(args[1]))
        elif op in ("s_add_u32", "s_addc_u32"):
            r = S(args[1]) + S(args[2]) + (w.scc if op == "s_addc_u32" else 0)
            w.scc = 1 if r > M32 else 0
            self.sset(w, args[0], r)
        elif op == "s_sub_u32":
            a, b = S(args[1]), S(args[2])
            w.scc = 1 if b > a else 0
            self.sset(w, args[0], a - b)
        elif op == "s_mul_i32":
            self.sset(w, args[0], S(args[1]) * S(args[2]))
        elif op == "s_mul_hi_u32":
            self.sset(w, args[0], (S(args[1]) * S(args[2])) >> 32)
        elif op == "s_lshl_b32":
            r = (S(args[1]) << (S(args[2]) & 31)) & M32
            w.scc = 1 if r else 0
            self.sset(w, args[0], r)
        elif op == "s_lshr_b32":
            r = S(args[1]) >> (S(args[2]) & 31)
            w.scc = 1 if r else 0
            self.sset(w, args[0], r)
        elif op == "s_and_b32":
            r = S(args[1]) & S(args[2])
            w.scc = 1 if r else 0
            self.sset(w, args[0], r)
        elif op == "s_cmp_eq_u32":
            w.scc = 1 if S(args[0]) == S(args[1]) else 0
        elif op == "s_cmp_lg_u32":
            w.scc = 1 if S(args[0]) != S(args[1]) else 0
        elif op == "s_cmp_ge_u32":
            w.scc = 1 if S(args[0]) >= S(args[1]) else 0
        elif op == "s_cselect_b64":
            lo = int(re.match(r"s\[(\d+):", args[0]).group(1))
            v = self.spair(w, args[1]) if w.scc else self.spair(w, args[2])
            w.s[lo], w.s[lo + 1] = v & M32, v >> 32
        elif op == "s_cselect_b32":
            self.sset(w, args[0], S(args[1]) if w.scc else S(args[2]))
        elif op == "s_branch":
            nxt = self.labels[args[0]]
        elif op == "s_cbranch_scc1":
            if w.scc:
                nxt = self.labels[args[0]]
        elif op in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mul_lo_u32", "v_sub_f32"):
            a, b = self.vsrc(w, args[1]), self.vsrc(w, args[2])
            if op == "v_and_b32": r = a & b
            elif op == "v_or_b32": r = a | b
            elif op == "v_xor_b32": r = a ^ b
            elif op == "v_add_u32": r = a + b
            elif op == "v_lshrrev_b32": r = b >> (a & 31)
            elif op == "v_lshlrev_b32": r = b << (a & 31)
            elif op == "v_mul_lo_u32": r = (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32)
            else: r = (a.view(np.float32) - b.view(np.float32)).view(np.uint32)
            w.v[self.rng(args[0])[1]] = r
        elif op == "v_bfe_u32":
            a, off, wd = self.vsrc(w, args[1]), int(args[2], 0), int(args[3], 0)
            w.v[self.rng(args[0])[1]] = (a >> off) & ((1 << wd) - 1)
        elif op == "v_lshl_add_u32":
            w.v[self.rng(args[0])[1]] = (self.vsrc(w, args[1]) << int(args[2], 0)) + self.vsrc(w, args[3])
        elif op == "v_mov_b32":
            w.v[self.rng(args[0])[1]] = self.vsrc(w, args[1])
        elif op == "v_readfirstlane_b32":
            self.sset(w, args[0], int(self.vsrc(w, args[1])[0]))
        elif op == "v_cmp_gt_u32":
            w.vcc = self.vsrc(w, args[1]) > self.vsrc(w, args[2])
        elif op == "v_cndmask_b32":
            w.v[self.rng(args[0])[1]] = np.where(w.vcc, self.vsrc(w, args[2]), self.vsrc(w, args[1]))
        elif op == "v_cvt_pk_f16_f32":
            lo = self.vsrc(w, args[1]).view(np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
            hi = self.vsrc(w, args[2]).view(np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
            w.v[self.rng(args[0])[1]] = lo | (hi << 16)
        elif op == "v_cvt_pk_bf16_f32":
            lo = f32_to_bf16(self.vsrc(w, args[1]).view(np.float32)).astype(np.uint32)
            hi = f32_to_bf16(self.vsrc(w, args[2]).view(np.float32)).astype(np.uint32)
            w.v[self.rng(args[0])[1]] = lo | (hi << 16)
        elif op == "v_cvt_f32_f16":
            w.v[self.rng(args[0])[1]] = (self.vsrc(w, args[1]) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32).view(np.uint32)
        elif op == "ds_read_b128":
            kind, lo, n = self.rng(args[0])
            m = re.fullmatch(r"v(\d+)(?:\s+offset:(\d+))?", args[1])
            addr = w.v[int(m.group(1))].astype(np.int64) + int(m.group(2) or 0)
            pieces = set()
            for l in range(64):
                a = int(addr[l])
                assert 0 <= a and a + 16 <= lds.size, f"LDS read out of range {a}"
                w.v[lo:lo + 4, l] = lds[a:a + 16].view(np.uint32)
                pieces.add(a >> 10)
            for p in pieces:
                st = lds_state.get(p)
                if st is None or not st.get("landed"):
                    self.errors.append(f"wave {w.wid} pc {w.pc} `{ins}`: reads LDS piece {p} that has not landed")
                elif st["writer"] != w.wid and not w.epoch > st["landed_epoch"]:
                    self.errors.append(f"wave {w.wid} pc {w.pc} `{ins}`: reads piece {p} of wave {st['writer']} without a barrier after its wait")
                if st is not None:
                    st.setdefault("readers", {})[w.wid] = w.epoch
            w.lgkm.append(pieces)
        elif op == "global_load_lds_dwordx4":
            voff = w.v[self.rng(args[0])[1]].astype(np.int64)
            base = self.spair(w, args[1])
            data = self.gload(base + voff, 16)
            dst = w.m0 & 0xffff_ffff
            assert dst % 1024 == 0 and dst + 1024 <= lds.size, f"LDS DMA destination {dst}"
            piece = dst >> 10
            st = lds_state.get(piece)
            if st is not None:
                for r, ep in st.get("readers", {}).items():
                    if r != w.wid and not ep < w.epoch:
                        self.errors.append(f"wave {w.wid} pc {w.pc}: overwrites LDS piece {piece} that wave {r} read in the same barrier epoch")
                if not st.get("landed"):
                    self.errors.append(f"wave {w.wid} pc {w.pc}: overwrites LDS piece {piece} whose previous load never retired")
            lds[dst:dst + 1024] = data.reshape(-1)
            self.seq = getattr(self, "seq", 0) + 1
            lds_state[piece] = {"writer": w.wid, "seq": self.seq, "landed": False, "readers": {}}
            w.vmq.append({"kind": "dma", "piece": piece, "seq": self.seq})
        elif op in ("global_load_dwordx4", "global_load_dword"):
            n = 4 if op.endswith("x4") else 1
            kind, lo, cnt = self.rng(args[0])
            assert cnt == n
            voff = w.v[self.rng(args[1])[1]].astype(np.int64)
            m = re.fullmatch(r"(s\[\d+:\d+\])(?:\s+offset:(\d+))?", args[2])
            base = self.spair(w, m.group(1)) + int(m.group(2) or 0)
            data = self.gload(base + voff, 4 * n).view(np.uint32)      # [64, n]
            rf = self.regfile(w, kind)
            rf[lo:lo + n] = data.T
            o = {"kind": "ld", "regs": [(kind, lo + i) for i in range(n)]}
            for r in o["regs"]:
                w.pending_reg[r] = o
            w.vmq.append(o)
        elif op == "global_store_dwordx4":
            voff = w.v[self.rng(args[0])[1]].astype(np.int64)
            kind, lo, cnt = self.rng(args[1])
            self.check_reg(w, kind, lo, cnt)
            m = re.fullmatch(r"(s\[\d+:\d+\])(?:\s+offset:(\d+))?", args[2])
            base = self.spair(w, m.group(1)) + int(m.group(2) or 0)
            rf = self.regfile(w, kind)
            for l in range(64):
                a = int(base + voff[l])
                assert 0 <= a and a + 16 <= self.mem.size
                self.mem[a:a + 16] = rf[lo:lo + 4, l].copy().view(np.uint8)
            w.vmq.append({"kind": "st"})
        elif op.startswith("v_mfma_f32_32x32x16"):
            self.stats["mfma"] += 1
            dk, dlo, dn = self.rng(args[0])
            ak, alo, _ = self.rng(args[1])
            bk, blo, _ = self.rng(args[2])
            self.check_reg(w, ak, alo, 4); self.check_reg(w, bk, blo, 4)
            if args[3] != "0":
                ck, clo, _ = self.rng(args[3])
                self.check_reg(w, ck, clo, 16)
                cin = self.regfile(w, ck)[clo:clo + 16].view(np.float32).copy()
            else:
                cin = np.zeros((16, 64), np.float32)

            def mat(kind, lo):
                regs = self.regfile(w, kind)[lo:lo + 4]            # [4 regs][64 lanes]
                h = np.zeros((64, 8), np.uint16)
                for j in range(4):
                    h[:, 2 * j] = regs[j] & 0xffff
                    h[:, 2 * j + 1] = regs[j] >> 16
                f = h.view(np.float16).astype(np.float32) if self.dt == "f16" else bf16_to_f32(h)
                m_ = np.zeros((32, 16), np.float32)
                for l in range(64):
                    m_[l & 31, (l >> 5) * 8:(l >> 5) * 8 + 8] = f[l]
                return m_
            A, B = mat(ak, alo), mat(bk, blo)
            D = A.astype(np.float64) @ B.astype(np.float64).T          # [row i][col j]
            out = cin.copy()
            for r in range(16):
                for hi in range(2):
                    i = (r & 3) + 8 * (r >> 2) + 4 * hi
                    out[r, hi * 32:hi * 32 + 32] += D[i, :].astype(np.float32)
            self.regfile(w, dk)[dlo:dlo + 16] = out.view(np.uint32)
        else:
            raise NotImplementedError(ins)
        w.pc = nxt
        if w.pc >= len(self.prog):
            w.done = True

    def run_wg(self, wg):
        lds = np.zeros(G.STAGE * G.NSTAGE, np.uint8)
        lds_state = {}
        waves = [Wave(i, wg) for i in range(G.NW)]
        for w in waves:
            w.s[2] = wg
        while not all(w.done for w in waves):
            for w in waves:
                while not w.done and not w.at_barrier:
                    self.step(w, lds, lds_state)
            if all(w.at_barrier or w.done for w in waves):
                if any(w.done for w in waves) and any(w.at_barrier for w in waves):
                    self.errors.append("a wave left while others wait at a barrier")
                    break
                for w in waves:
                    w.at_barrier = False
                    w.epoch += 1
        for w in waves:
            if w.vmq:
                self.errors.append(f"wave {w.wid} exits with {len(w.vmq)} VMEM operations in flight")


def build_tiles(tiles_m, tiles_n, grid):
    """the host-side tile lists of gemm_asm.hip (build_list)"""
    total, wpx, GROUP_M = tiles_m * tiles_n, grid >> 3, 8
    h = np.full((grid, G.TILE_SLOTS), 0xffffffff, np.uint32)
    for b in range(grid):
        xcd, idx = b & 7, b >> 3
        q, r = total >> 3, total & 7
        xs = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        cnt = q + (1 if xcd < r else 0)
        n = 0
        t = xs + idx
        while t < xs + cnt:
            per_group = GROUP_M * tiles_n
            gid = t // per_group
            first_m = gid * GROUP_M
            gsz = min(GROUP_M, tiles_m - first_m)
            rr = t - gid * per_group
            h[b, n] = ((first_m + rr % gsz) << 16) | (rr // gsz)
            n += 1
            t += wpx
    return h


def run(dt="f16", M=512, N=256, K=1024, grid=8, seed=0, verbose=True, mutate=None):
    """mutate: optional function(list of instruction lines) -> list, applied to the generated body (fault injection in the tests)"""
    rng = np.random.default_rng(seed)
    rows = (M + 255) // 256 * 256
    A = (rng.standard_normal((rows, K)) * 0.5).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = (rng.standard_normal((rows, N)) * 3).astype(np.float32)
    if dt == "f16":
        A16, W16 = A.astype(np.float16).view(np.uint16), W.astype(np.float16).view(np.uint16)
        Af, Wf = A16.view(np.float16).astype(np.float64), W16.view(np.float16).astype(np.float64)
    else:
        A16, W16 = f32_to_bf16(A), f32_to_bf16(W)
        Af, Wf = bf16_to_f32(A16).astype(np.float64), bf16_to_f32(W16).astype(np.float64)
    ref = C0.astype(np.float64) + Af @ Wf.T + bias
    tiles = build_tiles(rows // 256, N // 128, grid)
    # flat memory
    mem = np.zeros(64 << 20, np.uint8)
    cur = [4096]

    def put(arr):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        a = cur[0]
        mem[a:a + b.size] = b
        cur[0] = (a + b.size + 4095) // 4096 * 4096
        return a
    pA, pW, pC, pB, pT = put(A16), put(W16), put(C0), put(bias), put(tiles)
    pS = cur[0]; cur[0] += 256 * N * 4 + 8192
    karg = np.zeros(16, np.uint32)
    for i, p in enumerate((pA, pW, pC, pB, pT, pS)):
        karg[2 * i], karg[2 * i + 1] = p & M32, p >> 32
    karg[12], karg[13], karg[14], karg[15] = K // 64, K * 2, N * 4, 0
    pK = put(karg)
    lines = G.Gen(dt).kernel()
    if mutate is not None:
        lines = mutate(list(lines))
    emu = Emu(lines, dt, mem, pK, grid)
    for wg in range(grid):
        if (tiles[wg] != 0xffffffff).any() or True:
            emu.run_wg(wg)
    out = mem[pC:pC + rows * N * 4].view(np.float32).reshape(rows, N)
    err = np.abs(out[:M] - ref[:M]).max()
    scale = np.abs(ref).max()
    if verbose:
        print(f"{dt} M={M} N={N} K={K} grid={grid}: max|C - ref| = {err:.3e} (|ref| max {scale:.2f}); {emu.stats['mfma']} MFMAs, "
              f"{emu.stats['instr']} instructions; order-model findings: {len(emu.errors)}")
        for e_ in emu.errors[:12]:
            print("   ", e_)
    return err, scale, emu.errors


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--dt", default="f16"); ap.add_argument("--M", type=int, default=512); ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--K", type=int, default=1024); ap.add_argument("--grid", type=int, default=8)
    a = ap.parse_args()
    err, scale, errors = run(a.dt, a.M, a.N, a.K, a.grid)
    sys.exit(1 if errors or not err <= 2e-3 * scale else 0)
