#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 GEMM bodies (gemm_asm.hip includes the output, gemm_asm_body.inc).

Why a generator: the ViT block's residual GEMMs (attn.proj, mlp.fc2: timm Block invoked from lseg_vit.py:196-197) lose 20-47 % of a
launch in an epilogue that hipcc cannot overlap with the next tile's K-loop (three attempts in round 4 died in SIInsertWaitcnts /
register copies of in-flight loads).  Here the whole persistent loop is ONE asm statement: register allocation, every s_waitcnt count
and the instruction interleave are decided in this file, and the counts are CHECKED by the small queue / ring models below instead of
by reading the ISA afterwards.

Kernel (one workgroup = 4 waves = one per SIMD, 512 registers each, 144 KB LDS, persistent over a host-made tile list):
  tile 256 x 128, K-step 64; wave w owns rows 64w .. 64w+63 x all 128 columns = 2 x 4 blocks of v_mfma_f32_32x32x16 (weights as the row
  operand: a lane ends up with 4 consecutive output columns of one row -> 16-byte accesses);
  operands HBM -> LDS by global_load_lds_dwordx4 into a 3-stage ring (48 KB per stage: A 256 rows + W 128 rows of 128 bytes, XOR swizzle on
  the source address as in gemm.hip), fragments by ds_read_b128 one 16-deep sub-step ahead of their MFMAs;
  accumulators in the AGPR file, TWO sets of 128: while tile i accumulates into one set, the other set is stored (tile i-1) and then
  pre-loaded with the residual of tile i+1 -- C = res + A W^T needs no epilogue arithmetic at all when the residual IS the accumulator's
  start value (global_load / global_store move AGPRs directly); the bias rides in as one extra MFMA per block on a (hi, lo) 16-bit split of
  the fp32 bias against a ones operand.
  One s_barrier per K-step.  vmcnt is never 0 inside the loop.

Row padding contract: A, C / residual have ceil(M / 256) * 256 allocated rows (the engine pads its buffers); rows >= M are computed on
whatever the padding holds and never read by anyone.
"""
import sys

BM, BN, BK = 256, 128, 64
STAGE = (BM + BN) * 128            # 49152
NSTAGE = 3
W_OFF = BM * 128                   # W panel inside a stage
import os
NW = int(os.environ.get("GEMM_ASM_WAVES", "4"))   # 4: one wave per SIMD (64 x 128 per wave, 512 registers); 8: two per SIMD (64 x 64 per wave, 256)
WGN = NW // 4                      # waves along N (4 along M)
BNB = 4 // WGN                     # 32-column blocks per wave (2 row blocks of 32)
NBLK = 2 * BNB
ACCSET = 16 * NBLK                 # accumulator registers of one set
NPIECE_A, NPIECE_W = 32 // NW, 16 // NW          # 1 KB direct-to-LDS pieces per wave per K-step
NPIECE = NPIECE_A + NPIECE_W
TILE_SLOTS = 32                    # tile-list entries per workgroup (sentinel 0xffffffff)
DMA_SUBSTEPS = int(os.environ.get("GEMM_ASM_DMA_SUBSTEPS", "44"))     # the 12 pieces of a K-step go out over this many sub-steps behind the barrier (1, 2, 4)
READS_FRONT = int(os.environ.get("GEMM_ASM_READS_FRONT", "1"))        # fragment reads at the head of their sub-step (1) or spread over it (0)
ABL_NODMA = int(os.environ.get("GEMM_ASM_NODMA", "0"))                # attribution builds (tools/build_asm_variants.sh): no direct-to-LDS loads ...
ABL_NOREAD = int(os.environ.get("GEMM_ASM_NOREAD", "0"))              # ... no fragment reads (wrong results, timing only)
ABL_BARE = int(os.environ.get("GEMM_ASM_BARE", "0"))                  # ... nothing but MFMAs, waits and barriers in the K-steps
# sub-step (in issue order behind the barrier: 3, 0, 1, 2) -> piece indices.  Default 4 + 5 + 3: the cursor moves on in sub-step 1 behind the
# last piece, the fragment addresses in sub-step 2 -- no sub-step carries more than ~3.3 instructions per MFMA gap
_PLANS = {4: [4, 5, 3, 0], 44: [4, 4, 4, 0], 2: [6, 6, 0, 0], 3: [3, 3, 3, 3]} if NW == 4 else {4: [2, 3, 1, 0], 44: [2, 2, 2, 0], 2: [3, 3, 0, 0], 3: [2, 2, 1, 1]}
_cnt = _PLANS[DMA_SUBSTEPS]
DMA_PLAN, _j = {}, 0
for _ss, _n in zip([3, 0, 1, 2], _cnt):
    DMA_PLAN[_ss] = list(range(_j, _j + _n)); _j += _n
N_FIRST = len(DMA_PLAN[3])            # pieces of the NEXT group a K-step issues in its sub-step 3
U_BIAS = NBLK                      # K-step that loads the bias (behind NBLK / 2 steps of stores and NBLK / 2 of loads); + 2: convert; + 3: the bias MFMAs
EPI_UNITS = U_BIAS + 4             # unrolled K-steps at the head of a tile (carry the other accumulator set's stores / loads + the bias)

# ---- registers ------------------------------------------------------------------------------------------------------------------------
S_A, S_W, S_C, S_BIAS, S_TILES, S_SCR = 16, 18, 20, 22, 24, 26
S_NK, S_LDA, S_LDC, S_FLAGS = 28, 29, 30, 31
S_WAVE, S_TMP0, S_CSTAGE, S_DSTAGE, S_DLDS = 98, 33, 34, 35, 36     # (s32 is the ABI's stack pointer: left alone)
S_W1K = 37
S_DA, S_DW, S_DK = 38, 40, 42
S_NA, S_NW = 44, 46
S_CURC, S_PREVC, S_NEXTC = 48, 50, 52
S_CURB, S_NEXTB = 54, 56
S_LOOP = 58
S_T = 59                            # s59 .. s63 temporaries
S_ENT = 64                          # s64 .. s95 tile entries
S_KRING, S_TMP1 = 31, 99             # NSTAGE * STAGE (ring size); a temporary
S_WM, S_WN = 62, 63                  # the wave's place in the tile (prologue only: s62 / s63 are tile_bases temporaries afterwards; s100+ are reserved)
S_KFWD, S_KBACK = 96, 97            # +STAGE / -(NSTAGE - 1) * STAGE: fragment-address step to the next ring stage

V_LANE = 1
V_DSA, V_DSW = 2, 6                 # 4 + 4 fragment-read addresses (one per 16-deep sub-step)
V_DMA = 10                          # 12 per-lane source offsets
V_ROW = 22                          # 2 row offsets of the C / residual accesses (bm = 0, 1)
V_BOFF = 24                         # bias load offset
V_T = 25                            # v25 .. v31 temporaries
V_BFRAG = 32                        # 4 x 4 bias fragments
V_ONES = 48                         # 4
V_BRAW = 52                         # 4
V_FRAG = (64, 96)                   # fragment sets: A0 A1 W0 W1 W2 W3, 4 registers each


def acc(P, bm, bn):
    return ACCSET * P + 16 * (bm * BNB + bn)


class Gen:
    def __init__(self, dt):
        self.dt = dt                # "f16" | "bf16"
        self.out = []
        self.vq = []                # model of the in-order VMEM queue: one tag per issued operation
        self.label_n = 0

    def e(self, s):
        self.out.append(s)

    # -- VMEM bookkeeping --------------------------------------------------------------------------------------------------------------
    def vmem(self, text, tag):
        self.vq.append(tag)
        self.e(text)

    def wait_vm_for(self, pred, lgkm0=False, what=""):
        """s_waitcnt vmcnt(N): everything up to and including the YOUNGEST queued operation matching pred has completed."""
        idx = [i for i, t in enumerate(self.vq) if pred(t)]
        assert idx, "nothing to wait for: " + what
        n = len(self.vq) - 1 - idx[-1]
        assert n <= 63, n
        self.e(f"s_waitcnt vmcnt({n})" + (" lgkmcnt(0)" if lgkm0 else "") + (f"    ; {what}" if what else ""))
        return n

    # -- building blocks -----------------------------------------------------------------------------------------------------------------
    def mfma(self, a, wfrag, afrag, zero=False):
        c = "0" if zero else f"a[{a}:{a + 15}]"
        self.e(f"v_mfma_f32_32x32x16_{self.dt} a[{a}:{a + 15}], v[{wfrag}:{wfrag + 3}], v[{afrag}:{afrag + 3}], {c}")

    def frag_reads(self, fset, sub):
        """the 6 fragment reads of 16-deep sub-step `sub` (current ds address registers) into fragment set fset: A0 A1 W0..W3"""
        base = V_FRAG[fset]
        r = []
        for bm in range(2):
            r.append(f"ds_read_b128 v[{base + 4 * bm}:{base + 4 * bm + 3}], v{V_DSA + sub} offset:{bm * 4096}")
        for bn in range(BNB):
            r.append(f"ds_read_b128 v[{base + 8 + 4 * bn}:{base + 8 + 4 * bn + 3}], v{V_DSW + sub} offset:{bn * 4096}")
        return r

    def dma_piece(self, j, tag):
        """piece j of the K-step at the DMA cursor: (M0 write, pad, load).  Returns instruction list; the load is tagged when emitted.
        Queue tags of the pieces: "cur" = the step the cursor is on while a K-step's sub-steps 0-2 run (step k+2), "nxt" = the one its
        sub-step 3 starts (k+3), "prev" = k+1 (what the step's barrier waits for), "old" = landed."""
        if j < NPIECE_A:
            lds, src = j * NW * 1024, S_DA
        else:
            lds, src = W_OFF + (j - NPIECE_A) * NW * 1024, S_DW
        return (f"s_add_u32 m0, s{S_DLDS}, {lds}", f"global_load_lds_dwordx4 v{V_DMA + j}, s[{src}:{src + 1}]", tag)

    def cursor_advance(self):
        """the DMA cursor moves one K-step on (into the next tile behind a tile's last step) and to the next ring stage.  Atomic groups: an
        M0 write (which sets SCC) must not come between an SCC producer and its consumers."""
        g = []
        for s in (S_DA, S_DW):
            g.append([f"s_add_u32 s{s}, s{s}, 128", f"s_addc_u32 s{s + 1}, s{s + 1}, 0"])
        g.append([f"s_sub_u32 s{S_DK}, s{S_DK}, 1", f"s_cmp_eq_u32 s{S_DK}, 0",
                  f"s_cselect_b64 s[{S_DA}:{S_DA + 1}], s[{S_NA}:{S_NA + 1}], s[{S_DA}:{S_DA + 1}]",
                  f"s_cselect_b64 s[{S_DW}:{S_DW + 1}], s[{S_NW}:{S_NW + 1}], s[{S_DW}:{S_DW + 1}]",
                  f"s_cselect_b32 s{S_DK}, s{S_NK}, s{S_DK}"])
        g.append([f"s_add_u32 s{S_DLDS}, s{S_DLDS}, s{S_KFWD}", f"s_cmp_ge_u32 s{S_DLDS}, s{S_KRING}",
                  f"s_cselect_b32 s{S_TMP1}, s{S_KRING}, 0", f"s_sub_u32 s{S_DLDS}, s{S_DLDS}, s{S_TMP1}"])
        return [[("s", x) for x in grp] for grp in g]

    def stage_advance(self):
        """fragment-read addresses move to the next ring stage: issued behind the LAST reads of the current stage (sub-step 2's), ahead of the
        step's barrier"""
        g = [[("s", f"s_add_u32 s{S_CSTAGE}, s{S_CSTAGE}, 1"), ("s", f"s_cmp_eq_u32 s{S_CSTAGE}, {NSTAGE}"),
              ("s", f"s_cselect_b32 s{S_CSTAGE}, 0, s{S_CSTAGE}"), ("s", f"s_cselect_b32 s{S_TMP0}, s{S_KBACK}, s{S_KFWD}")]]
        for i in range(8):
            g.append([("s", f"v_add_u32 v{V_DSA + i}, s{S_TMP0}, v{V_DSA + i}")])
        return g

    def epi_ops(self, P, unit):
        """what the K-step `unit` of a tile on accumulator set P carries for the OTHER set Q (stores of tile i-1, then the residual of tile i+1)
        and for the tile's own bias.  Returns (list of filler instructions, list of extra MFMAs)."""
        Q = 1 - P
        ops, extra = [], []
        if unit < NBLK // 2:              # stores: blocks 2 unit, 2 unit + 1
            for b in (2 * unit, 2 * unit + 1):
                bm, bn = b // BNB, b % BNB
                for q in range(4):
                    a = acc(Q, bm, bn) + 4 * q
                    ops.append(("v", f"global_store_dwordx4 v{V_ROW + bm}, a[{a}:{a + 3}], s[{S_PREVC}:{S_PREVC + 1}] offset:{bn * 128 + q * 32}", "st"))
        elif unit < NBLK:
            for b in (2 * (unit - NBLK // 2), 2 * (unit - NBLK // 2) + 1):
                bm, bn = b // BNB, b % BNB
                for q in range(4):
                    a = acc(Q, bm, bn) + 4 * q
                    ops.append(("v", f"global_load_dwordx4 a[{a}:{a + 3}], v{V_ROW + bm}, s[{S_NEXTC}:{S_NEXTC + 1}] offset:{bn * 128 + q * 32}", "ld"))
        elif unit == U_BIAS:
            for bn in range(BNB):
                ops.append(("v", f"global_load_dword v{V_BRAW + bn}, v{V_BOFF}, s[{S_CURB}:{S_CURB + 1}] offset:{bn * 128}", "bias"))
        elif unit == U_BIAS + 2:          # two K-steps after the loads: the counted wait leaves the younger DMA pieces in flight
            ops.append(("w", "bias"))     # wait marker: resolved against the queue model when emitted
            for bn in range(BNB):
                b, t0, t1, f = V_BRAW + bn, V_T, V_T + 1, V_BFRAG + 4 * bn
                if self.dt == "f16":
                    grp = [("s", f"v_cvt_pk_f16_f32 v{t0}, v{b}, 0"), ("s", f"v_cvt_f32_f16 v{t0}, v{t0}"),
                           ("s", f"v_sub_f32 v{t1}, v{b}, v{t0}"), ("s", f"v_cvt_pk_f16_f32 v{t0}, v{b}, v{t1}")]
                else:
                    grp = [("s", f"v_cvt_pk_bf16_f32 v{t0}, v{b}, 0"), ("s", f"v_lshlrev_b32 v{t0}, 16, v{t0}"),
                           ("s", f"v_sub_f32 v{t1}, v{b}, v{t0}"), ("s", f"v_cvt_pk_bf16_f32 v{t0}, v{b}, v{t1}")]
                grp += [("s", f"v_cndmask_b32 v{f}, 0, v{t0}, vcc")]       # vcc = lanes 0 .. 31 (k-slots 0 .. 7), set once in the prologue
                ops.append(grp)           # one atomic group per chain: the chains share their temporaries
        elif unit == U_BIAS + 3:
            for b in range(NBLK):
                bm, bn = b // BNB, b % BNB
                extra.append((acc(P, bm, bn), V_BFRAG + 4 * bn, V_ONES))
        return ops, extra

    def emit_filler(self, op):
        if ABL_BARE and op[0] != "w":
            if op[0] == "v":
                self.vq.append(op[2])
            return
        if ABL_NODMA and op[0] == "v" and "global_load_lds" in op[1]:
            self.vq.append(op[2])          # the queue model keeps its counts (the emitted waits then pass at once)
            return
        if ABL_NOREAD and op[1].startswith("ds_read"):
            return
        if op[0] == "v":
            self.vmem(op[1], op[2])
        elif op[0] == "w":
            self.wait_vm_for(lambda t: t == op[1], what=f"{op[1]} loads have landed")
        else:
            self.e(op[1])

    def substep(self, P, fset, reads, pieces, groups, extra_mfma=(), groups_min_gap=0, pieces_dense=False):
        """The 8 MFMAs of one 16-deep sub-step on fragment set fset (+ the bias MFMAs when given) with everything else in the gaps behind them:
        `reads` two per gap from the first gap on; DMA `pieces` one per gap from gap 1 on -- the M0 write closes the gap BEFORE the load's,
        so an MFMA always separates the two; `groups` (atomic runs of other instructions, order kept) fill up evenly by instruction COUNT,
        none before gap groups_min_gap.  A wave that runs alone on its SIMD hides ~5 issue slots behind an MFMA, whatever their kind."""
        base = V_FRAG[fset]
        mf = []
        for bn in range(BNB):
            for bm in range(2):
                mf.append((acc(P, bm, bn), base + 8 + 4 * bn, base + 4 * bm))
        seq = []
        for i, m in enumerate(mf):
            seq.append(m)
            if extra_mfma:               # bias MFMA of the block 4 positions away (keeps same-accumulator MFMAs apart)
                bm, bn = i % 2, i // 2
                for x in extra_mfma:
                    if x[0] == acc(P, bm, (bn + BNB // 2) % BNB):
                        seq.append(x)
        nm = len(seq)
        gaps = [[] for _ in range(nm)]
        tail = [[] for _ in range(nm)]     # M0 writes: last in their gap
        r = list(reads)
        g = 0
        while r:
            gaps[g].append(("s", r.pop(0)))
            if r:
                gaps[g].append(("s", r.pop(0)))
            g += 1
        n = len(pieces)
        if n:
            stride = 1 if pieces_dense else max(1, (nm - 1) // n)
            for i, (m0w, load, tag) in enumerate(pieces):
                gp = 1 + i * stride
                assert gp < nm
                tail[gp - 1].append(("s", m0w))
                gaps[gp].insert(0, ("v", load, tag))
        flat = [x if isinstance(x, list) else [x] for x in groups]
        total = sum(len(x) for x in gaps) + sum(len(x) for x in tail) + sum(len(x) for x in flat)
        target = -(-total // nm)
        g = groups_min_gap
        for grp in flat:
            while g < nm - 1 and len(gaps[g]) + len(tail[g]) + len(grp) > max(target, len(grp)):
                g += 1
            gaps[g] += grp
        self.max_gap = max(getattr(self, "max_gap", 0), max(len(gaps[i]) + len(tail[i]) for i in range(nm)))
        for i, m in enumerate(seq):
            self.mfma(m[0], m[1], m[2])
            for f in gaps[i] + tail[i]:
                self.emit_filler(f)

    def unit(self, P, epi_unit, plain=False):
        """One K-step.  epi_unit: index of the step inside the tile for the epilogue schedule (None: carries nothing)."""
        ops, extra = self.epi_ops(P, epi_unit) if epi_unit is not None else ([], [])
        # split the epilogue operations over the four sub-steps (waits / conversions stay in order)
        quarter = [ops[i * len(ops) // 4:(i + 1) * len(ops) // 4] for i in range(4)] if ops else [[], [], [], []]
        if epi_unit == U_BIAS:
            quarter = [ops, [], [], []]        # the bias loads go out first
        for sub in range(4):
            fset = sub & 1
            plan = DMA_PLAN.get(sub, [])
            pieces = [self.dma_piece(j, "nxt" if sub == 3 else "cur") for j in plan]
            groups = list(quarter[sub])
            min_gap, dense = 0, False
            if (NPIECE - 1) in plan:           # the cursor moves on behind the group's last load
                groups = groups + self.cursor_advance()
                dense = True
                min_gap = len(plan)            # dense: piece i sits in gap 1 + i, the last one in gap len(plan)
            if sub == 2:                       # behind the stage's last fragment reads (gaps 0 .. 2)
                groups = groups + self.stage_advance()
                min_gap = max(min_gap, (2 + BNB + 1) // 2)
            if sub < 3:
                self.e("s_waitcnt lgkmcnt(0)")
                rd = self.frag_reads(1 - fset, sub + 1)
            else:
                # the step's barrier: my pieces of step k+1 have landed (only step k+2's 12 may still fly), every fragment read of this stage is back
                if plain:
                    assert not ops
                    self.e(f"s_waitcnt vmcnt({NPIECE}) lgkmcnt(0)")
                else:
                    self.wait_vm_for(lambda t: t == "prev", lgkm0=True, what="pieces of step k+1")
                self.e("s_barrier")
                rd = self.frag_reads(1 - fset, 0)      # next stage: the addresses moved on in sub-step 2
            self.substep(P, fset, rd, pieces, groups, extra if sub == 1 else (), min_gap, dense)
        self.age()

    def age(self):
        """end of a K-step: the queue tags move one step"""
        m = {"prev": "old", "cur": "prev", "nxt": "cur"}
        self.vq = [m.get(t, t) for t in self.vq]

    # -- whole kernel --------------------------------------------------------------------------------------------------------------------
    def tile_bases(self, ent, dA, dW, dC, dB, scratch_if_none, t):
        """SGPR pairs dA, dW, dC, dB <- bases of tile entry register `ent` (mb << 16 | nb).  scratch_if_none: entry 0xffffffff -> keep A / W of
        the CURRENT cursor tile and point C at the scratch tile.  t .. t+3: temporaries."""
        L = self.newlabel()
        o = []
        if scratch_if_none:
            o += [f"s_cmp_eq_u32 s{ent}, -1", f"s_cbranch_scc1 {L}_none"]
        o += [f"s_lshr_b32 s{t}, s{ent}, 16", f"s_and_b32 s{t + 1}, s{ent}, 0xffff",
              # A + mb * 256 * lda
              f"s_lshl_b32 s{t + 2}, s{S_LDA}, 8", f"s_mul_hi_u32 s{t + 3}, s{t}, s{t + 2}", f"s_mul_i32 s{t + 2}, s{t}, s{t + 2}",
              f"s_add_u32 s{dA}, s{S_A}, s{t + 2}", f"s_addc_u32 s{dA + 1}, s{S_A + 1}, s{t + 3}",
              # W + nb * 128 * lda
              f"s_lshl_b32 s{t + 2}, s{S_LDA}, 7", f"s_mul_hi_u32 s{t + 3}, s{t + 1}, s{t + 2}", f"s_mul_i32 s{t + 2}, s{t + 1}, s{t + 2}",
              f"s_add_u32 s{dW}, s{S_W}, s{t + 2}", f"s_addc_u32 s{dW + 1}, s{S_W + 1}, s{t + 3}",
              # C + mb * 256 * ldc + nb * 512
              f"s_lshl_b32 s{t + 2}, s{S_LDC}, 8", f"s_mul_hi_u32 s{t + 3}, s{t}, s{t + 2}", f"s_mul_i32 s{t + 2}, s{t}, s{t + 2}",
              f"s_add_u32 s{dC}, s{S_C}, s{t + 2}", f"s_addc_u32 s{dC + 1}, s{S_C + 1}, s{t + 3}",
              f"s_lshl_b32 s{t + 2}, s{t + 1}, 9", f"s_add_u32 s{dC}, s{dC}, s{t + 2}", f"s_addc_u32 s{dC + 1}, s{dC + 1}, 0",
              f"s_add_u32 s{dB}, s{S_BIAS}, s{t + 2}", f"s_addc_u32 s{dB + 1}, s{S_BIAS + 1}, 0"]
        if scratch_if_none:
            o += [f"s_branch {L}_done", f"{L}_none:",
                  f"s_mov_b32 s{dC}, s{S_SCR}", f"s_mov_b32 s{dC + 1}, s{S_SCR + 1}",
                  f"s_mov_b32 s{dB}, s{S_BIAS}", f"s_mov_b32 s{dB + 1}, s{S_BIAS + 1}",
                  f"{L}_done:"]
        for x in o:
            self.e(x)

    def newlabel(self):
        self.label_n += 1
        return f".Lg{self.dt}_{self.label_n}_%="

    def prologue(self):
        e = self.e
        e(f"s_load_dwordx8 s[{S_A}:{S_A + 7}], %0, 0x0")
        e(f"s_load_dwordx8 s[{S_TILES}:{S_TILES + 7}], %0, 0x20")
        e(f"v_and_b32 v{V_LANE}, 63, %2")
        e(f"v_lshrrev_b32 v{V_T}, 6, %2")
        e("s_nop 0")
        e(f"v_readfirstlane_b32 s{S_WAVE}, v{V_T}")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 4")
        # this workgroup's tile list
        e(f"s_lshl_b32 s{S_T}, %1, {TILE_SLOTS.bit_length() - 1 + 2}")
        e(f"s_add_u32 s{S_TILES}, s{S_TILES}, s{S_T}")
        e(f"s_addc_u32 s{S_TILES + 1}, s{S_TILES + 1}, 0")
        e(f"s_load_dwordx16 s[{S_ENT}:{S_ENT + 15}], s[{S_TILES}:{S_TILES + 1}], 0x0")
        e(f"s_load_dwordx16 s[{S_ENT + 16}:{S_ENT + 31}], s[{S_TILES}:{S_TILES + 1}], 0x40")
        e(f"s_lshl_b32 s{S_W1K}, s{S_WAVE}, 10")
        e(f"s_mov_b32 s{S_KRING}, {NSTAGE * STAGE}")
        e(f"s_mov_b32 s{S_KFWD}, {STAGE}")
        e(f"s_mov_b32 s{S_KBACK}, {-(NSTAGE - 1) * STAGE & 0xffffffff:#x}")
        # ---- lane constants
        # l31 = lane & 31, hi = lane >> 5, swl = (l31 >> 1) & 7
        e(f"v_and_b32 v{V_T}, 31, v{V_LANE}")                         # l31
        e(f"v_lshrrev_b32 v{V_T + 1}, 5, v{V_LANE}")                  # hi
        e(f"v_bfe_u32 v{V_T + 2}, v{V_LANE}, 1, 3")                   # swl
        e(f"v_lshlrev_b32 v{V_T + 3}, 7, v{V_T}")                     # l31 * 128
        e(f"s_lshr_b32 s{S_WM}, s{S_WAVE}, {WGN.bit_length() - 1}")   # wave (wm, wn): rows 64 wm .., columns 32 BNB wn ..
        e(f"s_and_b32 s{S_WN}, s{S_WAVE}, {WGN - 1}")
        e(f"s_lshl_b32 s{S_T}, s{S_WM}, 13")                          # wave's A rows: 64 wm * 128 bytes
        e(f"s_mul_i32 s{S_T + 1}, s{S_WN}, {BNB * 4096}")             # wave's W rows
        for s in range(4):
            # chunk = (2 s | hi) ^ swl
            e(f"v_or_b32 v{V_T + 4}, {2 * s}, v{V_T + 1}")
            e(f"v_xor_b32 v{V_T + 4}, v{V_T + 4}, v{V_T + 2}")
            e(f"v_lshl_add_u32 v{V_DSW + s}, v{V_T + 4}, 4, v{V_T + 3}")
            e(f"v_add_u32 v{V_DSA + s}, s{S_T}, v{V_DSW + s}")
            e(f"v_add_u32 v{V_DSW + s}, {W_OFF}, v{V_DSW + s}")
            e(f"v_add_u32 v{V_DSW + s}, s{S_T + 1}, v{V_DSW + s}")
        # DMA source offsets: piece i of A covers slab (w + 4 i): rows 8 (w + 4 i) + (lane >> 3); slot lane & 7 holds chunk (lane & 7) ^ swz(row),
        # swz(row) = (row >> 1) & 7 = (4 (w + 4 i) + (lane >> 4)) & 7 = (4 (w & 1) + (lane >> 4)) & 7
        e(f"v_lshrrev_b32 v{V_T}, 3, v{V_LANE}")                      # lane >> 3
        e(f"v_lshrrev_b32 v{V_T + 1}, 4, v{V_LANE}")                  # lane >> 4
        e(f"s_and_b32 s{S_T}, s{S_WAVE}, 1")
        e(f"s_lshl_b32 s{S_T}, s{S_T}, 2")
        e(f"v_add_u32 v{V_T + 1}, s{S_T}, v{V_T + 1}")
        e(f"v_and_b32 v{V_T + 1}, 7, v{V_T + 1}")                     # swz
        e(f"v_and_b32 v{V_T + 2}, 7, v{V_LANE}")
        e(f"v_xor_b32 v{V_T + 2}, v{V_T + 2}, v{V_T + 1}")            # source chunk
        e(f"v_lshlrev_b32 v{V_T + 2}, 4, v{V_T + 2}")                 # * 16 bytes
        e(f"s_lshl_b32 s{S_T}, s{S_WAVE}, 3")
        e(f"v_add_u32 v{V_T}, s{S_T}, v{V_T}")                        # row of piece 0: 8 w + (lane >> 3)
        e(f"v_mul_lo_u32 v{V_T}, v{V_T}, s{S_LDA}")
        e(f"v_add_u32 v{V_T}, v{V_T}, v{V_T + 2}")                    # piece 0 offset (same formula for A and W: rows relative to the panel)
        e(f"s_lshl_b32 s{S_T}, s{S_LDA}, {(NW * 8).bit_length() - 1}")   # 8 NW rows per piece index
        for j in range(NPIECE):
            i = j if j < NPIECE_A else j - NPIECE_A
            if i == 0:
                e(f"v_mov_b32 v{V_DMA + j}, v{V_T}")
            else:
                e(f"v_add_u32 v{V_DMA + j}, s{S_T}, v{V_DMA + j - 1}")
        # C / residual row offsets: (64 w + 32 bm + l31) * ldc + hi * 16
        e(f"v_and_b32 v{V_T}, 31, v{V_LANE}")
        e(f"s_lshl_b32 s{S_T}, s{S_WM}, 6")
        e(f"v_add_u32 v{V_T}, s{S_T}, v{V_T}")
        e(f"v_lshrrev_b32 v{V_T + 1}, 5, v{V_LANE}")
        e(f"v_lshlrev_b32 v{V_T + 1}, 4, v{V_T + 1}")
        e(f"s_mul_i32 s{S_T + 1}, s{S_WN}, {BNB * 128}")              # the wave's first column, in bytes of fp32
        e(f"v_add_u32 v{V_T + 1}, s{S_T + 1}, v{V_T + 1}")
        e(f"v_mul_lo_u32 v{V_ROW}, v{V_T}, s{S_LDC}")
        e(f"v_add_u32 v{V_ROW}, v{V_ROW}, v{V_T + 1}")
        e(f"s_lshl_b32 s{S_T}, s{S_LDC}, 5")
        e(f"v_add_u32 v{V_ROW + 1}, s{S_T}, v{V_ROW}")
        # bias: lane loads bias[32 bn + l31]; fragments live in k-slots 0, 1 of lanes 0 .. 31 only
        e(f"v_and_b32 v{V_BOFF}, 31, v{V_LANE}")
        e(f"v_lshlrev_b32 v{V_BOFF}, 2, v{V_BOFF}")
        e(f"v_add_u32 v{V_BOFF}, s{S_T + 1}, v{V_BOFF}")
        e(f"v_cmp_gt_u32 vcc, 32, v{V_LANE}")
        one2 = 0x3C003C00 if self.dt == "f16" else 0x3F803F80
        e(f"v_mov_b32 v{V_T}, {one2:#x}")
        e(f"v_cndmask_b32 v{V_ONES}, 0, v{V_T}, vcc")
        for k in range(1, 4):
            e(f"v_mov_b32 v{V_ONES + k}, 0")
        for k in range(4 * BNB):
            e(f"v_mov_b32 v{V_BFRAG + k}, 0")
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_cmp_eq_u32 s{S_ENT}, -1")          # a workgroup without tiles (grid rounded up to a multiple of 8) leaves before the first barrier
        e("s_cbranch_scc1 .Lexit_%=")
        # ---- state: current tile = entry 0
        e(f"s_mov_b32 s{S_CSTAGE}, 0")
        e(f"s_mov_b32 s{S_DSTAGE}, 0")
        e(f"s_mov_b32 s{S_DLDS}, s{S_W1K}")
        e(f"s_mov_b32 s{S_DK}, s{S_NK}")
        self.tile_bases(S_ENT, S_DA, S_DW, S_CURC, S_CURB, False, S_T)
        e(f"s_mov_b32 s{S_PREVC}, s{S_SCR}")
        e(f"s_mov_b32 s{S_PREVC + 1}, s{S_SCR + 1}")
        # cursor's "next tile" for the prologue's own crossings: none can happen (nk >= 16 checked on the host); still keep it defined
        for k in range(2):
            e(f"s_mov_b32 s{S_NA + k}, s{S_DA + k}")
            e(f"s_mov_b32 s{S_NW + k}, s{S_DW + k}")
        # residual of tile 0 -> accumulator set 0
        for b in range(NBLK):
            bm, bn = b // BNB, b % BNB
            for q in range(4):
                a = acc(0, bm, bn) + 4 * q
                self.vmem(f"global_load_dwordx4 a[{a}:{a + 3}], v{V_ROW + bm}, s[{S_CURC}:{S_CURC + 1}] offset:{bn * 128 + q * 32}", "ld")
        # K-steps 0, 1 and the first three pieces of step 2
        def piece(j, tag):
            m0w, load, t = self.dma_piece(j, tag)
            self.e(m0w)
            self.e("s_nop 0")
            self.emit_filler(("v", load, t))

        def advance():
            for grp in self.cursor_advance():
                for op in grp:
                    self.e(op[1])
        for step in range(2):
            for j in range(NPIECE):
                piece(j, f"s{step}")
            advance()
        for j in range(N_FIRST):
            piece(j, "s2")
        if N_FIRST == NPIECE:
            advance()
        # step 0 has landed (step 1 and the first pieces of step 2 may fly)
        n = self.wait_vm_for(lambda t: t == "s0", what="residual of tile 0 and K-step 0")
        assert n == NPIECE + N_FIRST
        e("s_barrier")
        for x in self.frag_reads(0, 0):
            e(x)

    def tile_head(self):
        """top of a tile body: bases of the NEXT tile (cursor crossing, residual pre-load); entry S_ENT = this tile, S_ENT + 1 = next"""
        # A / W of the next tile: when there is none the cursor keeps re-issuing the current tile (never consumed, keeps the counts exact)
        L = self.newlabel()
        e = self.e
        e(f"s_cmp_eq_u32 s{S_ENT + 1}, -1")
        e(f"s_cselect_b32 s{S_T + 4}, s{S_ENT}, s{S_ENT + 1}")
        # A / W from s(T+4); C / bias from the real next entry (scratch when none)
        self.tile_bases(S_T + 4, S_NA, S_NW, S_NEXTC, S_NEXTB, False, S_T)
        e(f"s_cmp_eq_u32 s{S_ENT + 1}, -1")
        e(f"s_cselect_b32 s{S_NEXTC}, s{S_SCR}, s{S_NEXTC}")
        e(f"s_cselect_b32 s{S_NEXTC + 1}, s{S_SCR + 1}, s{S_NEXTC + 1}")

    def tile_tail(self, P):
        """bottom of a tile body: rotate the C pointers and the tile list; leave to the other body or to the drain"""
        e = self.e
        for k in range(2):
            e(f"s_mov_b32 s{S_PREVC + k}, s{S_CURC + k}")
            e(f"s_mov_b32 s{S_CURC + k}, s{S_NEXTC + k}")
            e(f"s_mov_b32 s{S_CURB + k}, s{S_NEXTB + k}")
        for k in range(TILE_SLOTS - 1):
            e(f"s_mov_b32 s{S_ENT + k}, s{S_ENT + k + 1}")
        e(f"s_mov_b32 s{S_ENT + TILE_SLOTS - 1}, -1")
        e(f"s_cmp_eq_u32 s{S_ENT}, -1")
        e(f"s_cbranch_scc1 .Ldrain{P}_%=")

    def body(self, P):
        e = self.e
        e(f".Lbody{P}_%=:")
        self.tile_head()
        # the queue model at a tile's first step: older steps' pieces only
        self.vq = ["old"] * 8 + ["prev"] * NPIECE + ["cur"] * N_FIRST
        for u in range(EPI_UNITS):
            e(f"; ---- tile step {u} (accumulator set {P})")
            self.unit(P, u)
        # every epilogue operation of this tile is older than a "dma_prev" piece by now: the plain step's vmcnt(12) covers them
        assert all(t in ("prev", "cur") for t in self.vq[-(NPIECE + N_FIRST):]), self.vq[-30:]
        e(f"s_sub_u32 s{S_LOOP}, s{S_NK}, {EPI_UNITS}")
        e(f"s_cmp_eq_u32 s{S_LOOP}, 0")
        e(f"s_cbranch_scc1 .Lend{P}_%=")
        e(f".Lloop{P}_%=:")
        self.unit(P, None, plain=True)
        e(f"s_sub_u32 s{S_LOOP}, s{S_LOOP}, 1")
        e(f"s_cmp_lg_u32 s{S_LOOP}, 0")
        e(f"s_cbranch_scc1 .Lloop{P}_%=")
        e(f".Lend{P}_%=:")
        self.tile_tail(P)
        e(f"s_branch .Lbody{1 - P}_%=")

    def drain(self, P):
        e = self.e
        e(f".Ldrain{P}_%=:")
        e("s_nop 15")
        e("s_nop 7")
        for b in range(NBLK):
            bm, bn = b // BNB, b % BNB
            for q in range(4):
                a = acc(P, bm, bn) + 4 * q
                e(f"global_store_dwordx4 v{V_ROW + bm}, a[{a}:{a + 3}], s[{S_PREVC}:{S_PREVC + 1}] offset:{bn * 128 + q * 32}")
        e("s_branch .Lexit_%=")

    def kernel(self):
        self.prologue()
        self.e(f"s_branch .Lbody0_%=")
        self.body(0)
        self.body(1)
        self.drain(0)
        self.drain(1)
        self.e(".Lexit_%=:")
        self.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        return self.out


def check_first_use(lines):
    """cheap structural checks on the emitted text: every s_waitcnt vmcnt inside the bodies is counted (never 0), M0 writes are padded."""
    inloop = False
    for i, l in enumerate(lines):
        if l.startswith(".Lbody"):
            inloop = True
        if l.startswith(".Ldrain"):
            inloop = False
        if inloop and "s_waitcnt vmcnt(0)" in l:
            raise SystemExit(f"vmcnt(0) inside the tile loop at line {i}: {l}")
        if l.startswith("s_add_u32 m0") and not (ABL_NODMA or ABL_BARE):
            nxt = [x for x in lines[i + 1:i + 8] if not x.startswith("v_mfma")]
            k = next(j for j, x in enumerate(lines[i + 1:i + 12]) if x.startswith("global_load_lds"))
            assert k >= 1, (i, lines[i:i + 4])                        # at least one instruction between the M0 write and the load that reads it
            assert not any(x.startswith("s_add_u32 m0") for x in lines[i + 1:i + 1 + k]), (i, lines[i:i + 6])


def main():
    out_path = sys.argv[1]
    tmp_path = out_path + ".tmp"           # a failing run must not leave a half-written include behind
    with open(tmp_path, "w") as f:
        f.write("// generated by gemm_asm_gen.py -- do not edit\n")
        for dt in ("f16", "bf16"):
            g = Gen(dt)
            lines = g.kernel()
            check_first_use(lines)
            f.write(f"#define LSEG_GEMM_ASM_BODY_{dt.upper()} \\\n")
            for l in lines:
                assert '"' not in l
                f.write('    "' + l.split("    ;")[0].rstrip() + '\\n" \\\n')
            f.write('    ""\n\n')
            n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
            print(f"{dt}: {len(lines)} lines, {n_mfma} MFMAs, fullest MFMA gap {g.max_gap} instructions", file=sys.stderr)
        # registers the bodies own: named as clobbers so that the compiler keeps its operands out of them and the kernel descriptor
        # allocates the whole file (512 registers per lane, one wave per SIMD)
        cl = [f"v{i}" for i in range(1, 128)] + [f"a{i}" for i in range(2 * ACCSET)] + [f"s{i}" for i in range(16, 100) if i != 32] + ["vcc", "scc", "memory"]
        f.write("#define LSEG_GEMM_ASM_CLOBBERS " + ", ".join('"' + c + '"' for c in cl) + "\n")
        f.write(f"#define LSEG_GEMM_ASM_TILE_SLOTS {TILE_SLOTS}\n#define LSEG_GEMM_ASM_LDS {STAGE * NSTAGE}\n#define LSEG_GEMM_ASM_MIN_KSTEPS {EPI_UNITS + 4}\n"
                f"#define LSEG_GEMM_ASM_THREADS {64 * NW}\n#define LSEG_GEMM_ASM_WAVES_PER_EU {NW // 4}\n")
    os.replace(tmp_path, out_path)


if __name__ == "__main__":
    main()
