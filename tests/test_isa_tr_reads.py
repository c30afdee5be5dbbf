"""The K-major weight-gradient GEMM reads its MFMA fragments with INLINE-ASM `ds_read_b64_tr_b16` that hipcc's s_waitcnt insertion does not
see, and waits `lgkmcnt(0)` itself before the MFMA block that consumes them (csrc/gemm.hip, ds_read_tr16_pair_asm).  Whether that is
correct depends on the register allocation of THIS build: any compiler-generated instruction that touches a destination VGPR between a read
and the wait (a v_mov assembling the 128-bit operand, a spill, a copy of a loop-carried fragment) would read registers the LDS has not
filled yet (VERDICT r4 "exactly the kind of change that produces a read-before-land race").  Checked on the disassembly of the
shipped liblseg_hip.so along every control-flow path -- no GPU needed -- so a compiler bump or an edit that breaks it fails the CPU suite."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_tr_read_check as ISA                                            # noqa: E402


@pytest.mark.skipif(not os.path.exists(ISA.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_nothing_touches_an_asm_transpose_read_destination_before_its_lgkmcnt0():
    assert os.path.exists(ISA.DEFAULT_LIB), "liblseg_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    nk, nr, report = ISA.check_library(ISA.DEFAULT_LIB)
    assert nk >= 2 and nr >= 64, (nk, nr)                                  # bf16 + fp16 instances of the 128x128 K-major kernel, 48 reads each
    assert not report, {k: v[:4] for k, v in report.items()}


def test_the_checker_itself_flags_a_hazard():
    """A synthetic function: the second read's destination is consumed by a v_mov before the wait."""
    ins = [(0, "ds_read_b64_tr_b16", "v[4:5], v2 offset:64", None),
           (8, "ds_read_b64_tr_b16", "v[6:7], v2 offset:128", None),
           (16, "v_mfma_f32_16x16x32_bf16", "v[20:23], v[10:13], v[14:17], v[20:23]", None),
           (24, "v_mov_b32_e32", "v9, v6", None),
           (28, "s_waitcnt", "lgkmcnt(0)", None),
           (32, "v_mfma_f32_16x16x32_bf16", "v[20:23], v[4:7], v[14:17], v[20:23]", None),
           (40, "s_endpgm", "", None)]
    bad = ISA.check_function(ins)
    assert len(bad) == 1 and bad[0][0] == 8 and bad[0][1] == 24
    ok = [i for i in ins if i[0] != 24]
    assert ISA.check_function(ok) == []
    # ... and along a branch: the wait sits on the fall-through path only
    br = [(0, "ds_read_b64_tr_b16", "v[4:5], v2", None), (8, "s_cbranch_scc1", "3", 24), (12, "s_waitcnt", "lgkmcnt(0)", None),
          (16, "s_branch", "2", 32), (24, "v_add_u32_e32", "v4, v4, v1", None), (28, "s_waitcnt", "lgkmcnt(0)", None), (32, "s_endpgm", "", None)]
    assert [b[1] for b in ISA.check_function(br)] == [24]
