"""The hand-scheduled residual GEMM (lang-seg_amd/csrc/gemm_asm_gen.py -> gemm_asm.hip) on the CPU: tools/asm_gemm_emu.py executes the
generated instruction text lane by lane (4 waves, SGPR / VGPR / AGPR files, LDS ring, flat memory) and replays the ordering rules the
hardware gives (in-order VMEM queue + counted vmcnt, barriers, direct-to-LDS pieces).  What it pins:
  * C = C + A W^T + bias (attn.proj / mlp.fc2 of a timm Block, lseg_vit.py:196-197) on ragged M, several tiles per workgroup, both bodies,
    the plain-step loop and the drain, fp16 and bf16;
  * every s_waitcnt count of the schedule: a deliberately loosened count MUST be reported by the order model."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("dt,M,N,K", [("f16", 1500, 384, 1280), ("bf16", 700, 128, 1024)])
def test_generated_body_computes_the_residual_gemm(dt, M, N, K):
    import asm_gemm_emu as E
    err, scale, findings = E.run(dt, M, N, K, grid=8, verbose=False)
    assert not findings, findings[:5]
    assert err <= 2e-6 * scale * (K / 64) ** 0.5 + 1e-5, (err, scale)        # fp32 accumulation of 16-bit products vs fp64


def test_order_model_catches_a_loosened_wait():
    """fault injection: the plain K-step's barrier wait lets 24 instead of 12 operations fly -> fragment reads of pieces that have not landed"""
    import asm_gemm_emu as E

    def loosen(lines):
        hits = [i for i, l in enumerate(lines) if l.startswith("s_waitcnt vmcnt(12) lgkmcnt(0)")]
        assert hits
        for i in hits:
            lines[i] = "s_waitcnt vmcnt(24) lgkmcnt(0)"
        return lines
    _, _, findings = E.run("f16", 512, 128, 4096, grid=8, verbose=False, mutate=loosen)
    assert any("has not landed" in f for f in findings), findings[:3]


def test_order_model_catches_a_missing_barrier():
    import asm_gemm_emu as E

    def drop(lines):
        hits = [i for i, l in enumerate(lines) if l == "s_barrier"]
        lines[hits[len(hits) // 2]] = "s_nop 0"          # one barrier of the unrolled head of a tile
        return lines
    _, _, findings = E.run("f16", 512, 128, 1024, grid=8, verbose=False, mutate=drop)
    assert findings, "a dropped barrier went unnoticed"
