"""Per-kernel parity through the C ABI (liblseg_hip.so) against plain fp32 torch on the
same inputs.  Operands are rounded to the kernel's operand type first, so the only
difference left is accumulation order (fp32) and the output rounding."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.gpu_fast]

from lseg_hip import _lib  # noqa: E402

DT = {torch.float32: _lib.LSEG_F32, torch.float16: _lib.LSEG_F16, torch.bfloat16: _lib.LSEG_BF16}


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _lib.load()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(901, 1024, 1024), (256, 128, 64), (130, 192, 128), (3604, 3072, 1024),
                                   (77, 512, 2048), (5, 64, 64)])
def test_gemm_plain(lib, dtype, M, N, K):
    A = rnd((M, K), dtype, 1)
    W = rnd((N, K), dtype, 2, 1 / math.sqrt(K))
    out = torch.full((M, N), float("nan"), dtype=torch.float32).cuda()
    _lib.check(lib.lseg_op_gemm(P(A), P(W), None, None, P(out), M, N, K, DT[dtype], _lib.LSEG_F32, 0, stream()))
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t()
    err = (out - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogues(lib, act):
    M, N, K = 333, 256, 192
    dtype = torch.bfloat16
    A = rnd((M, K), dtype, 3)
    W = rnd((N, K), dtype, 4, 1 / math.sqrt(K))
    bias = rnd((N,), torch.float32, 5)
    res = rnd((M, N), torch.float32, 6)
    ref = A.float() @ W.float().t() + bias
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    elif act == 3:
        ref = F.relu(ref)
    ref = ref + res
    for od, tol in ((torch.float32, 2e-3), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)):
        out = torch.zeros((M, N), dtype=od).cuda()
        _lib.check(lib.lseg_op_gemm(P(A), P(W), P(bias), P(res), P(out), M, N, K, DT[dtype], DT[od], act, stream()))
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item()
        assert err < tol * max(1.0, ref.abs().max().item()), (od, err)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,ntok,max_grid", [(2, 901, 0), (9, 901, 0), (3, 197, 0), (9, 901, 16)])
def test_vit_block_gemm_forms(lib, dtype, B, ntok, max_grid):
    """The four GEMMs of a timm Block on their specialised epilogues (lseg_op_gemm_vit): Linear, Linear + GELU(erf), the fp32 residual
    read-modify-write, and the qkv Linear with timm's reshape(B,N,3,H,D/H).permute folded into the store (q, k head-major, v
    transposed) -- against fp32 torch on the rounded operands.  B = 9 runs the 256 x 256 tile configuration (8109 rows), max_grid = 16
    the persistent schedule with many tiles per workgroup."""
    D, H = 1024, 16
    M, npad = B * ntok, (ntok + 127) // 128 * 128
    A = rnd((M, D), dtype, 11)
    tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3

    def run(kind, N, K, Ain, Cq, Ck=None, Cv=None, seed=12):
        W = rnd((N, K), dtype, seed, 1 / math.sqrt(K))
        bias = rnd((N,), torch.float32, seed + 1)
        _lib.check(lib.lseg_op_gemm_vit(P(Ain), P(W), P(bias), P(Cq), P(Ck), P(Cv), M, N, K, DT[dtype], kind, ntok, npad, max_grid, stream()))
        torch.cuda.synchronize()
        return Ain.float() @ W.float().t() + bias

    out = torch.zeros((M, D), dtype=dtype).cuda()
    ref = run(0, D, D, A, out)
    assert (out.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    hid = torch.zeros((M, 4 * D), dtype=dtype).cuda()
    ref = F.gelu(run(1, 4 * D, D, A, hid, seed=14))
    assert (hid.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    x0 = rnd((M, D), torch.float32, 16, 3.0)
    x = x0.clone()
    ref = x0 + run(2, D, 4 * D, hid, x, seed=18)
    assert (x - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    q = torch.full((B * H, npad, 64), 7.0, dtype=dtype).cuda()
    k = torch.full((B * H, npad, 64), 7.0, dtype=dtype).cuda()
    vt = torch.full((B * H, 64, npad), 7.0, dtype=dtype).cuda()
    ref = run(3, 3 * D, D, A, q, k, vt, seed=20).reshape(B, ntok, 3, H, 64).permute(2, 0, 3, 1, 4).reshape(3, B * H, ntok, 64)
    big = max(1.0, ref.abs().max().item())
    assert (q[:, :ntok].float() - ref[0]).abs().max().item() < tol * big
    assert (k[:, :ntok].float() - ref[1]).abs().max().item() < tol * big
    assert (vt[:, :, :ntok].float() - ref[2].transpose(1, 2)).abs().max().item() < tol * big
    # pad rows / columns are not written
    assert (q[:, ntok:] == 7).all() and (k[:, ntok:] == 7).all() and (vt[:, :, ntok:] == 7).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(256, 128, 1024), (700, 256, 1024), (1500, 384, 1280), (2 * 901, 1024, 4096), (9 * 901, 1024, 1024)])
def test_hand_scheduled_residual_gemm(lib, dtype, M, N, K):
    """C (fp32, in place) += A W^T + bias -- attn.proj / mlp.fc2 of a timm Block (lseg_vit.py:196-197) -- on the hand-scheduled 256 x 128
    kernel (csrc/gemm_asm.hip, impl = 1) against fp64 torch on the rounded operands and against the generic kernel family (impl = 0):
    ragged M (the buffers reach the next multiple of 256 rows, the contract of rows_alloc), one to several tiles per workgroup, K = 1024
    (the unrolled head of a tile only) and longer (the plain-step loop)."""
    rows = (M + 255) // 256 * 256
    A = rnd((rows, K), dtype, 31)
    W = rnd((N, K), dtype, 32, 1 / math.sqrt(K))
    bias = rnd((N,), torch.float32, 33)
    C0 = rnd((rows, N), torch.float32, 34, 3.0)
    ref = C0[:M].double() + A[:M].double() @ W.double().t() + bias.double()
    out = {}
    for impl in (1, 0):
        Cb = C0.clone()
        _lib.check(lib.lseg_op_gemm_res32(P(A), P(W), P(bias), P(Cb), M, N, K, rows, DT[dtype], impl, 0, stream()))
        torch.cuda.synchronize()
        out[impl] = Cb
    big = max(1.0, ref.abs().max().item())
    e_asm = (out[1][:M].double() - ref).abs().max().item()
    e_gen = (out[0][:M].double() - ref).abs().max().item()
    # fp32 accumulation of exact 16-bit products: both kernels sit at fp32 round-off of the result (the hand-scheduled one accumulates ON the
    # residual, the generic one adds it last: a few ulp of |C| apart)
    assert e_gen < 1e-5 * big and e_asm < 2e-5 * big * max(1.0, (K / 1024) ** 0.5), (e_asm, e_gen, big)
    assert torch.isfinite(out[1]).all()


def test_hand_scheduled_residual_gemm_refuses_unpadded_buffers(lib):
    """rows_alloc below the next multiple of 256: the kernel would touch rows that do not exist -- impl = 1 must refuse, impl = -1 falls back"""
    M, N, K = 300, 128, 1024
    A, W, bias = rnd((M, K), torch.float16, 1), rnd((N, K), torch.float16, 2, 0.03), rnd((N,), torch.float32, 3)
    Cb = rnd((M, N), torch.float32, 4)
    ref = Cb + A.float() @ W.float().t() + bias
    assert lib.lseg_op_gemm_res32(P(A), P(W), P(bias), P(Cb), M, N, K, M, DT[torch.float16], 1, 0, stream()) == -5      # LSEG_ERR_UNSUPPORTED
    _lib.check(lib.lseg_op_gemm_res32(P(A), P(W), P(bias), P(Cb), M, N, K, M, DT[torch.float16], -1, 0, stream()))
    torch.cuda.synchronize()
    assert (Cb - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("R,C_,ld,off", [(7208, 1024, 1024, 0), (901, 513, 520, 0), (3333, 96, 96, 3), (130, 8, 8, 1), (60000, 256, 256, 0)])
@pytest.mark.parametrize("det", [False, True])
def test_column_sums_atomic_and_deterministic(lib, dtype, R, C_, ld, off, det):
    """The bias-gradient column sum (lseg_op_colsum) against fp64, ragged R / C, a leading dimension above C, an output that is not 16-byte
    aligned (`off` floats into its buffer) -- on the fp32-atomics path and on the DETERMINISTIC path (partial rows + fixed-order reduction:
    lseg_config.flags bit 3, what the training step runs by default), which must also repeat bit for bit and honour `accumulate`."""
    x = rnd((R, ld), dtype, 41)
    ref = x[:, :C_].double().sum(0)
    scale = x[:, :C_].double().abs().sum(0).max().item()
    ws = torch.empty(1 << 20, dtype=torch.float32, device="cuda") if det else None
    buf = torch.full((C_ + 8,), 7.0, dtype=torch.float32, device="cuda")
    out = buf[off:off + C_]
    outs = []
    for rep in range(2):
        _lib.check(lib.lseg_op_colsum(P(x), DT[dtype], P(out), R, C_, ld, 0, P(ws), ws.numel() if det else 0, stream()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert (outs[0].double() - ref).abs().max().item() <= 2e-6 * scale + 1e-6, (outs[0].double() - ref).abs().max().item()
    assert (buf[:off] == 7).all() and (buf[off + C_:] == 7).all()          # nothing outside the C columns is touched
    if det:
        assert torch.equal(outs[0], outs[1])                                 # fixed summation order
    _lib.check(lib.lseg_op_colsum(P(x), DT[dtype], P(out), R, C_, ld, 1, P(ws), ws.numel() if det else 0, stream()))
    torch.cuda.synchronize()
    assert (out.double() - 2 * ref).abs().max().item() <= 4e-6 * scale + 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,C_", [(2, 30, 30, 256), (1, 15, 17, 64), (8, 60, 60, 256)])
@pytest.mark.parametrize("det", [False, True])
def test_batchnorm_batch_sums_atomic_and_deterministic(lib, dtype, B, H, W, C_, det):
    """BatchNorm batch statistics (sum, sum of squares) and the two sums of its backward over padded NHWC maps (train-mode
    ResidualConvUnit_custom, lseg_blocks.py:276-283) against fp64, atomics and deterministic form; the deterministic form repeats bit for bit."""
    xi = torch.randn((B, C_, H, W), generator=torch.Generator().manual_seed(51)) * 2 + 0.5
    dyi = torch.randn((B, C_, H, W), generator=torch.Generator().manual_seed(52))
    x, dy = _pad_nhwc(xi.cuda(), dtype), _pad_nhwc(dyi.cuda(), dtype)
    xr = x[:, 1:-1, 1:-1, :].double().reshape(-1, C_)
    dr = dy[:, 1:-1, 1:-1, :].double().reshape(-1, C_)
    n = xr.shape[0]
    ws = torch.empty(1 << 20, dtype=torch.float32, device="cuda") if det else None
    cap = ws.numel() if det else 0
    stats = torch.empty(2 * C_, dtype=torch.float32, device="cuda")
    bst = torch.empty(2 * C_, dtype=torch.float32, device="cuda")
    runs = []
    for rep in range(2):
        _lib.check(lib.lseg_op_bn_stats(P(x), P(stats), B, H, W, C_, DT[dtype], P(ws), cap, stream()))
        _lib.check(lib.lseg_op_bn_bwd_stats(P(dy), P(x), P(stats), P(bst), B, H, W, C_, 1e-5, DT[dtype], P(ws), cap, stream()))
        torch.cuda.synchronize()
        runs.append((stats.clone(), bst.clone()))
    s1, s2 = xr.sum(0), (xr * xr).sum(0)
    assert (runs[0][0][:C_].double() - s1).abs().max().item() <= 2e-6 * xr.abs().sum(0).max().item()
    assert (runs[0][0][C_:].double() - s2).abs().max().item() <= 2e-6 * s2.max().item()
    mean = s1 / n
    rstd = ((s2 / n - mean * mean).clamp_min(0) + 1e-5).rsqrt()
    b1, b2 = dr.sum(0), (dr * (xr - mean) * rstd).sum(0)
    assert (runs[0][1][:C_].double() - b1).abs().max().item() <= 2e-6 * dr.abs().sum(0).max().item()
    assert (runs[0][1][C_:].double() - b2).abs().max().item() <= 2e-5 * (dr.abs() * ((xr - mean) * rstd).abs()).sum(0).max().item()
    if det:
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_gemm_transpose_detecting(lib):
    """A = I (padded), asymmetric W: catches a swapped C layout (cdna guide G9)."""
    M = N = K = 64
    A = torch.eye(M, K).to(torch.bfloat16).cuda()
    W = (torch.arange(N * K).reshape(N, K).float() % 251 / 16).to(torch.bfloat16).cuda()
    out = torch.zeros((M, N), dtype=torch.float32).cuda()
    _lib.check(lib.lseg_op_gemm(P(A), P(W), None, None, P(out), M, N, K, _lib.LSEG_BF16, _lib.LSEG_F32, 0, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, W.float().t())


@pytest.mark.parametrize("in_dt,out_dt,D,eps", [(torch.float32, torch.bfloat16, 1024, 1e-6),
                                                (torch.float16, torch.float16, 512, 1e-5),
                                                (torch.float32, torch.float16, 128, 1e-6),
                                                (torch.float32, torch.bfloat16, 768, 1e-6)])
def test_layernorm(lib, in_dt, out_dt, D, eps):
    M = 515
    x = rnd((M, D), in_dt, 7, 2.0) + 0.5
    g = (1 + 0.1 * rnd((D,), torch.float32, 8)).contiguous()
    b = (0.1 * rnd((D,), torch.float32, 9)).contiguous()
    out = torch.zeros((M, D), dtype=out_dt).cuda()
    _lib.check(lib.lseg_op_layernorm(P(x), DT[in_dt], P(g), P(b), P(out), DT[out_dt], M, D, eps, stream()))
    torch.cuda.synchronize()
    ref = F.layer_norm(x.float(), (D,), g, b, eps)
    tol = 2e-2 if out_dt == torch.bfloat16 else 3e-3
    assert (out.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def _attention_case(lib, dtype, B, H, N, causal, seed, spike=False):
    Npad = ((N + 127) // 128) * 128
    q = rnd((B, H, N, 64), dtype, seed)
    k = rnd((B, H, N, 64), dtype, seed + 1)
    v = rnd((B, H, N, 64), dtype, seed + 2)
    if spike:   # one key dominates one query late in the sequence -> running max jumps (online-softmax rescale)
        k[:, :, N - 3] = q[:, :, 5] * 4.0
    qp = torch.zeros((B * H, Npad, 64), dtype=dtype).cuda(); qp[:, :N] = q.reshape(B * H, N, 64)
    kp = torch.zeros((B * H, Npad, 64), dtype=dtype).cuda(); kp[:, :N] = k.reshape(B * H, N, 64)
    vt = torch.zeros((B * H, 64, Npad), dtype=dtype).cuda(); vt[:, :, :N] = v.reshape(B * H, N, 64).transpose(1, 2)
    out = torch.full((B, N, H * 64), float("nan"), dtype=dtype).cuda()
    _lib.check(lib.lseg_op_attention(P(qp), P(kp), P(vt), P(out), B, H, N, Npad, DT[dtype], int(causal), 0.125, stream()))
    torch.cuda.synchronize()
    s = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((N, N), float("-inf"), device=s.device).triu_(1)
    ref = (s.softmax(-1) @ v.float()).transpose(1, 2).reshape(B, N, H * 64)
    err = (out.float() - ref).abs().max().item()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert math.isfinite(err) and err < tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,N,causal", [(1, 2, 901, False), (2, 3, 77, True), (1, 1, 64, False),
                                          (1, 2, 130, True), (3, 16, 226, False)])
def test_attention(lib, dtype, B, H, N, causal):
    _attention_case(lib, dtype, B, H, N, causal, 20)


def test_attention_rescale_branch(lib):
    _attention_case(lib, torch.bfloat16, 1, 2, 901, False, 30, spike=True)
    _attention_case(lib, torch.float16, 1, 2, 77, True, 31, spike=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,N,mode", [(1, 2, 901, "plain"), (3, 16, 226, "plain"), (1, 2, 901, "spike"), (1, 2, 901, "ramp"),
                                        (1, 2, 901, "low"), (40, 16, 197, "plain")])
def test_attention_prescaled(lib, dtype, B, H, N, mode):
    """The inference engine's attention: q arrives as T(q * scale * log2 e) and the kernel bakes the reference max into the QK^T
    accumulator (attention.hip PRE = true; the reference max moves only when a tile exceeds it by 2^8).  Against fp32 torch on the same
    rounded operands.  spike: one key dominates one query late in the sequence (the refresh branch fires mid-sequence); ramp: scores
    grow steadily along the keys (refresh after refresh); low: every score far BELOW zero (the first-tile reference must be the tile's
    own max, not 0, or every probability underflows); B = 40 runs the 4-wave kernel on a shape whose last query block is mostly padding."""
    Npad = ((N + 127) // 128) * 128
    c = 0.125 * 1.4426950408889634
    q32 = torch.randn((B, H, N, 64), generator=torch.Generator().manual_seed(60)).cuda()
    k = rnd((B, H, N, 64), dtype, 61)
    v = rnd((B, H, N, 64), dtype, 62)
    if mode == "spike":
        k[:, :, N - 3] = (q32[:, :, 5] * 6.0).to(dtype)
    elif mode == "ramp":
        k = (k.float() + q32.mean(dim=2, keepdim=True) * torch.linspace(0, 24, N, device="cuda").view(1, 1, N, 1)).to(dtype)
        q32 = q32 + q32.mean(dim=2, keepdim=True) * 3
    elif mode == "low":
        k = (k.float() * 0.05 - q32.mean(dim=2, keepdim=True) * 30).to(dtype)
        q32 = q32 * 0.05 + q32.mean(dim=2, keepdim=True) * 8
    q = (q32 * c).to(dtype)                                        # ONE rounding, as the QKV epilogue does it
    qp = torch.zeros((B * H, Npad, 64), dtype=dtype).cuda(); qp[:, :N] = q.reshape(B * H, N, 64)
    kp = torch.zeros((B * H, Npad, 64), dtype=dtype).cuda(); kp[:, :N] = k.reshape(B * H, N, 64)
    vt = torch.zeros((B * H, 64, Npad), dtype=dtype).cuda(); vt[:, :, :N] = v.reshape(B * H, N, 64).transpose(1, 2)
    out = torch.full((B, N, H * 64), float("nan"), dtype=dtype).cuda()
    lse2 = torch.full((B * H, Npad), float("nan"), dtype=torch.float32).cuda()
    _lib.check(lib.lseg_op_attention_prescaled(P(qp), P(kp), P(vt), P(out), P(lse2), B, H, N, Npad, DT[dtype], stream()))
    torch.cuda.synchronize()
    s2 = q.double() @ k.double().transpose(-1, -2)                 # log2 units
    m = s2.amax(-1, keepdim=True)
    p = torch.exp2(s2 - m)
    ref = ((p / p.sum(-1, keepdim=True)) @ v.double()).transpose(1, 2).reshape(B, N, H * 64).float()
    ref_lse = (m.squeeze(-1) + torch.log2(p.sum(-1))).reshape(B * H, N).float()
    err = (out.float() - ref).abs().max().item()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert math.isfinite(err) and err < tol * max(1.0, ref.abs().max().item()), (mode, err)
    lerr = (lse2[:, :N] - ref_lse).abs().max().item()
    assert math.isfinite(lerr) and lerr < (3e-2 if dtype == torch.bfloat16 else 5e-3), (mode, lerr)


def _pad_nhwc(x_nchw, dtype):
    B, Cc, H, W = x_nchw.shape
    p = torch.zeros((B, H + 2, W + 2, Cc), dtype=dtype, device=x_nchw.device)
    p[:, 1:-1, 1:-1] = x_nchw.permute(0, 2, 3, 1).to(dtype)
    return p.contiguous()


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,relu_in,relu_out,with_res",
                         [(1, 15, 15, 64, 64, 1, 0, 0, False), (2, 30, 30, 128, 64, 1, 1, 1, False),
                          (1, 30, 30, 64, 128, 2, 0, 0, False), (1, 24, 16, 256, 256, 1, 1, 0, True),
                          (1, 120, 120, 256, 256, 1, 0, 0, True)])
def test_conv3x3(lib, B, H, W, Cin, Cout, stride, relu_in, relu_out, with_res):
    dt = torch.bfloat16
    x = rnd((B, Cin, H, W), dt, 40)
    w = rnd((Cout, Cin, 3, 3), dt, 41, 1 / math.sqrt(9 * Cin))
    bias = rnd((Cout,), torch.float32, 42)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rnd((B, Cout, Ho, Wo), dt, 43) if with_res else None
    xp = _pad_nhwc(x, dt)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()     # [Co, (ky,kx), Ci]
    resp = _pad_nhwc(res, dt) if with_res else None
    out = torch.zeros((B, Ho + 2, Wo + 2, Cout), dtype=dt).cuda()
    _lib.check(lib.lseg_op_conv3x3(P(xp), P(wp), P(bias), P(resp), P(out), B, H, W, Cin, Cout, stride,
                                   relu_in, relu_out, stream()))
    torch.cuda.synchronize()
    xin = F.relu(x.float()) if relu_in else x.float()
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1)
    if relu_out:
        ref = F.relu(ref)
    if with_res:
        ref = ref + res.float()
    got = out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
    assert (got - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    # the zero border must be untouched
    assert out[:, 0].abs().max().item() == 0 and out[:, :, 0].abs().max().item() == 0
    assert out[:, -1].abs().max().item() == 0 and out[:, :, -1].abs().max().item() == 0


def test_upsample2x(lib):
    dt = torch.bfloat16
    x = rnd((2, 64, 15, 10), dt, 50)
    out = torch.zeros((2, 30, 20, 64), dtype=dt).cuda()
    _lib.check(lib.lseg_op_upsample2x_nhwc(P(_pad_nhwc(x, dt)), P(out), 2, 15, 10, 64, stream()))
    ref = F.interpolate(x.float(), scale_factor=2, mode="bilinear", align_corners=True)
    assert (out.permute(0, 3, 1, 2).float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item()
    pl = rnd((7, 24, 20), torch.float32, 51)
    o2 = torch.zeros((7, 48, 40), dtype=torch.float32).cuda()
    _lib.check(lib.lseg_op_upsample2x_planes(P(pl), P(o2), 7, 24, 20, stream()))
    torch.cuda.synchronize()
    ref2 = F.interpolate(pl[None], scale_factor=2, mode="bilinear", align_corners=True)[0]
    assert (o2 - ref2).abs().max().item() < 1e-5


@pytest.mark.parametrize("K", [150, 7, 2, 1000])
def test_correlation_matches_oracle(lib, K):
    """lseg_net.py:187-196 as restated in oracle.correlate: bit-exact except where the fp32
    accumulation order changes an fp16 rounding (allow 1 fp16 ulp on <1% of logits)."""
    from oracle.lseg_oracle import correlate, LOGIT_SCALE, r16
    B, Pp, Cc = 2, 1000, 512
    g = torch.Generator().manual_seed(60 + K)
    feat = torch.randn((B * Pp, Cc), generator=g) * 3
    text = torch.randn((K, Cc), generator=g).half()
    ref = correlate(feat, text.float(), LOGIT_SCALE)                        # [M, K]
    tn = r16(text.float() / r16(text.float().norm(dim=-1, keepdim=True))).half().cuda()
    out = torch.zeros((B, K, Pp), dtype=torch.float32).cuda()
    _lib.check(lib.lseg_op_correlation(P(feat.cuda()), P(tn), P(out), B, Pp, Cc, K, LOGIT_SCALE, stream()))
    torch.cuda.synchronize()
    got = out.permute(0, 2, 1).reshape(B * Pp, K).cpu()
    assert torch.equal(got, r16(got))                                       # fp16-representable
    diff = (got - ref).abs()
    assert diff.max().item() <= 2 ** -6                                      # <= 1 ulp at |logit| < 16
    assert (diff > 0).float().mean().item() < 0.02
    assert (got.argmax(1) != ref.argmax(1)).float().mean().item() < 0.01


@pytest.mark.parametrize("M,Fd", [(1000, 256), (57600, 256), (130, 64)])
def test_fused_head_features(lib, M, Fd):
    """head1 + L2 norm + fp16 casts in one GEMM epilogue == the unfused reference arithmetic."""
    from oracle.lseg_oracle import LOGIT_SCALE, r16
    dt = torch.bfloat16
    x = rnd((M, Fd), dt, 70, 2.0)
    w = rnd((512, Fd), dt, 71, 1 / math.sqrt(Fd))
    b = rnd((512,), torch.float32, 72)
    out = torch.zeros((M, 512), dtype=torch.float16).cuda()
    _lib.check(lib.lseg_op_head_features(P(x), P(w), P(b), P(out), M, Fd, LOGIT_SCALE, stream()))
    torch.cuda.synchronize()
    v = x.float() @ w.float().t() + b
    ref = r16(LOGIT_SCALE * r16(v / v.norm(dim=-1, keepdim=True)))
    d = (out.float() - ref).abs()
    assert d.max().item() <= 2 ** -7                      # 1 fp16 ulp at |a| < 8 (fp32 summation order)
    assert (d > 0).float().mean().item() < 0.05


@pytest.mark.parametrize("dtype,M,N,K", [(torch.bfloat16, 901, 1024, 1024), (torch.float16, 300, 128, 256),
                                         (torch.bfloat16, 64, 64, 64), (torch.bfloat16, 1802, 4096, 1024),
                                         (torch.bfloat16, 130, 192, 320), (torch.bfloat16, 1000, 192, 256), (torch.float16, 37, 320, 128),
                                         (torch.bfloat16, 7208, 1024, 4096)])
def test_linear_backward(lib, dtype, M, N, K):
    """Backward of y = x W^T + b (first brick of the training step): dX = dY W, dW = dY^T X, db = sum_m dY -- against
    fp32 torch on the same 16-bit-rounded operands (what autograd computes under LSegmentationModule.training_step)."""
    dy = rnd((M, N), dtype, 1, 0.5)
    x = rnd((M, K), dtype, 2)
    w = rnd((N, K), dtype, 3, 1.0 / math.sqrt(K))
    dx = torch.empty((M, K), dtype=dtype, device="cuda")
    dw = torch.empty((N, K), dtype=torch.float32, device="cuda")
    db = torch.empty((N,), dtype=torch.float32, device="cuda")
    _lib.check(lib.lseg_op_linear_backward(P(dy), P(x), P(w), DT[dtype], P(dx), P(dw), P(db), M, N, K, stream()))
    torch.cuda.synchronize()
    dyf, xf, wf = dy.float(), x.float(), w.float()
    ref_dx, ref_dw, ref_db = dyf @ wf, dyf.t() @ xf, dyf.sum(0)
    eps = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    assert (dx.float() - ref_dx).abs().max().item() <= 2 * eps * ref_dx.abs().max().item() + 1e-6
    assert (dw - ref_dw).abs().max().item() <= 2e-4 * ref_dw.abs().max().item() + 1e-6      # fp32 accumulation order only
    assert (db - ref_db).abs().max().item() <= 2e-4 * max(1.0, ref_db.abs().max().item())
    # partial calls: any output may be omitted
    dw2 = torch.zeros_like(dw)
    _lib.check(lib.lseg_op_linear_backward(P(dy), P(x), P(w), DT[dtype], None, P(dw2), None, M, N, K, stream()))
    torch.cuda.synchronize()
    assert torch.equal(dw2, dw)


@pytest.mark.parametrize("dy_dtype,M,D", [(torch.float32, 901, 1024), (torch.bfloat16, 3604, 1024), (torch.float32, 37, 768),
                                           (torch.bfloat16, 130, 128)])
def test_layernorm_backward(lib, dy_dtype, M, D):
    """LayerNorm backward (timm norm1/norm2, eps 1e-6) against torch autograd in fp32 on the same inputs; dx also in
    accumulate mode (the residual stream's gradient receives the LN branch on top of the skip path)."""
    x = rnd((M, D), torch.float32, 1, 2.0) + 0.5
    gamma = (1.0 + 0.1 * rnd((D,), torch.float32, 2)).contiguous()
    beta = 0.1 * rnd((D,), torch.float32, 3)
    dy = rnd((M, D), dy_dtype, 4)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    F.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy.float())
    dx = torch.empty_like(x)
    dg, dbt = torch.empty_like(gamma), torch.empty_like(beta)
    _lib.check(lib.lseg_op_layernorm_backward(P(dy), DT[dy_dtype], P(x), P(gamma), P(dx), P(dg), P(dbt), M, D, 1e-6, 0, stream()))
    torch.cuda.synchronize()
    assert (dx - xr.grad).abs().max().item() <= 2e-5 * max(1.0, xr.grad.abs().max().item())
    assert (dg - gr.grad).abs().max().item() <= 2e-4 * max(1.0, gr.grad.abs().max().item())
    assert (dbt - br.grad).abs().max().item() <= 2e-4 * max(1.0, br.grad.abs().max().item())
    acc = torch.ones_like(x)
    _lib.check(lib.lseg_op_layernorm_backward(P(dy), DT[dy_dtype], P(x), P(gamma), P(acc), P(dg), P(dbt), M, D, 1e-6, 1, stream()))
    torch.cuda.synchronize()
    assert (acc - (1.0 + xr.grad)).abs().max().item() <= 2e-5 * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 15, 15, 64, 64), (1, 30, 30, 128, 256), (1, 12, 20, 256, 64), (2, 30, 30, 256, 256)])
def test_conv3x3_backward(lib, B, H, W, Cin, Cout):
    """Backward of the padded-NHWC 3x3 conv (stride 1, no bias): dX through the forward implicit-GEMM kernel with
    flipped / channel-swapped weights, dW as one GEMM over nine row-shifted transposes -- against torch autograd."""
    dt = torch.bfloat16
    x = rnd((B, Cin, H, W), dt, 50)
    w = rnd((Cout, Cin, 3, 3), dt, 51, 1 / math.sqrt(9 * Cin))
    dy = rnd((B, Cout, H, W), dt, 52)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=1, padding=1).backward(dy.float())
    xp, dyp = _pad_nhwc(x, dt), _pad_nhwc(dy, dt)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    dxp = torch.zeros((B, H + 2, W + 2, Cin), dtype=dt).cuda()
    dw = torch.empty((Cout, 9 * Cin), dtype=torch.float32).cuda()
    _lib.check(lib.lseg_op_conv3x3_backward(P(dyp), P(xp), P(wp), P(dxp), P(dw), B, H, W, Cin, Cout, stream()))
    torch.cuda.synchronize()
    got_dx = dxp[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
    assert (got_dx - xr.grad).abs().max().item() <= 2e-2 * max(1.0, xr.grad.abs().max().item())
    ref_dw = wr.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    assert (dw - ref_dw).abs().max().item() <= 2e-3 * max(1.0, ref_dw.abs().max().item())
    assert dxp[:, 0].abs().max().item() == 0 and dxp[:, :, -1].abs().max().item() == 0       # border stays zero


@pytest.mark.parametrize("quick", [False, True])
def test_gelu_backward(lib, quick):
    """erf GELU (timm Mlp) and CLIP's QuickGELU x*sigmoid(1.702x)."""
    dt = torch.float16 if quick else torch.bfloat16
    pre, dy = rnd((4099,), dt, 60, 2.0), rnd((4099,), dt, 61)
    pr = pre.float().requires_grad_(True)
    (pr * torch.sigmoid(1.702 * pr) if quick else F.gelu(pr)).backward(dy.float())
    dx = torch.empty_like(pre)
    fn = lib.lseg_op_quickgelu_backward if quick else lib.lseg_op_gelu_backward
    _lib.check(fn(P(dy), P(pre), P(dx), 4099, DT[dt], stream()))
    torch.cuda.synchronize()
    assert (dx.float() - pr.grad).abs().max().item() <= 2 ** -7 * max(1.0, pr.grad.abs().max().item())


@pytest.mark.parametrize("B,H,W,Cc", [(2, 15, 15, 64), (1, 8, 20, 256), (1, 2, 2, 8)])
def test_upsample2x_nhwc_backward(lib, B, H, W, Cc):
    """Transpose of the x2 bilinear (align_corners=True) NHWC upsample against torch autograd."""
    dt = torch.bfloat16
    dout = rnd((B, Cc, 2 * H, 2 * W), dt, 62)
    xr = torch.zeros((B, Cc, H, W), device="cuda", requires_grad=True)
    F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True).backward(dout.float())
    din = torch.zeros((B, H + 2, W + 2, Cc), dtype=dt).cuda()
    dout_nhwc = dout.permute(0, 2, 3, 1).contiguous()
    _lib.check(lib.lseg_op_upsample2x_nhwc_backward(P(dout_nhwc), P(din), B, H, W, Cc, stream()))
    torch.cuda.synchronize()
    got = din[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
    assert (got - xr.grad).abs().max().item() <= 2e-2 * max(1.0, xr.grad.abs().max().item())
    assert din[:, 0].abs().max().item() == 0 and din[:, :, 0].abs().max().item() == 0


@pytest.mark.parametrize("B,K,H,W", [(2, 7, 24, 20), (1, 150, 32, 32)])
def test_softmax_ce_backward(lib, B, K, H, W):
    """d logits of CrossEntropyLoss(ignore_index=-1) (mean over valid pixels), chained behind lseg_op_seg_stats."""
    g = torch.Generator().manual_seed(K)
    scores = (torch.randn((B, K, H, W), generator=g) * 3).cuda()
    target = torch.randint(0, K, (B, H, W), generator=g)
    target[torch.rand((B, H, W), generator=g) < 0.25] = -1
    target = target.cuda()
    sr = scores.clone().requires_grad_(True)
    F.cross_entropy(sr, target, ignore_index=-1).backward()
    counts = torch.empty(2 + 3 * K, dtype=torch.int64, device="cuda")
    nll = torch.empty(2, dtype=torch.float64, device="cuda")
    _lib.check(lib.lseg_op_seg_stats(P(scores), P(target), B, K, H, W, -1, P(counts), P(nll), stream()))
    dz = torch.empty_like(scores)
    _lib.check(lib.lseg_op_softmax_ce_backward(P(scores), P(target), P(dz), B, K, H, W, -1, P(nll), stream()))
    torch.cuda.synchronize()
    assert (dz - sr.grad).abs().max().item() <= 1e-5 * max(1e-3, sr.grad.abs().max().item()) + 1e-9
    assert dz[(target < 0).unsqueeze(1).expand_as(dz)].abs().max().item() == 0


@pytest.mark.parametrize("dtype,B,H,N", [(torch.bfloat16, 1, 2, 130), (torch.float16, 2, 3, 64), (torch.bfloat16, 1, 2, 901),
                                          (torch.bfloat16, 2, 1, 37), (torch.bfloat16, 3, 4, 257)])
def test_attention_backward_qkv(lib, dtype, B, H, N):
    """The attention backward the training step runs (csrc/attention_bwd2.hip: dQ kernel + dK/dV kernel, no atomics, output written
    straight as d(qkv Linear output) [B*N, 3*H*64]) against torch autograd in fp32 on the same 16-bit-rounded q, k, v."""
    Npad = ((N + 127) // 128) * 128
    D = H * 64
    q, k, v = rnd((B, H, N, 64), dtype, 70), rnd((B, H, N, 64), dtype, 71), rnd((B, H, N, 64), dtype, 72)
    d_o = rnd((B, N, D), dtype, 73)
    qp = torch.zeros((B * H, Npad, 64), dtype=dtype).cuda(); qp[:, :N] = q.reshape(B * H, N, 64)
    kp = torch.zeros((B * H, Npad, 64), dtype=dtype).cuda(); kp[:, :N] = k.reshape(B * H, N, 64)
    vt = torch.zeros((B * H, 64, Npad), dtype=dtype).cuda(); vt[:, :, :N] = v.reshape(B * H, N, 64).transpose(1, 2)
    out = torch.zeros((B, N, D), dtype=dtype).cuda()
    _lib.check(lib.lseg_op_attention(P(qp), P(kp), P(vt), P(out), B, H, N, Npad, DT[dtype], 0, 0.125, stream()))
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    s = (qr @ kr.transpose(-1, -2)) * 0.125
    ref = (s.softmax(-1) @ vr).transpose(1, 2).reshape(B, N, D)
    ref.backward(d_o.float())
    lse2 = torch.zeros((B * H, Npad), dtype=torch.float32).cuda()
    lse2[:, :N] = (torch.logsumexp(s.detach(), dim=-1) * 1.4426950408889634).reshape(B * H, N)
    dqkv = torch.full((B * N, 3 * D), float("nan"), dtype=dtype).cuda()
    assert lib.lseg_op_attention_backward_ws_bytes(B, H, Npad) >= 5 * B * H * Npad * 64 * 2
    _lib.check(lib.lseg_op_attention_backward_qkv(P(qp), P(kp), P(vt), P(out), P(d_o), P(lse2), P(dqkv), None, B, H, N, Npad,
                                                  DT[dtype], 0.125, stream()))
    torch.cuda.synchronize()
    got = dqkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)             # [3, B, H, N, 64]
    tol = 3e-2 if dtype == torch.bfloat16 else 6e-3
    for i, (want, name) in enumerate(((qr.grad, "dq"), (kr.grad, "dk"), (vr.grad, "dv"))):
        err = (got[i] - want).abs().max().item()
        rel = ((got[i] - want).norm() / want.norm()).item()
        assert math.isfinite(err) and err <= tol * max(1.0, want.abs().max().item()) and rel <= tol, (name, err, rel)


@pytest.mark.parametrize("B,H,W,Cc", [(2, 15, 15, 64), (1, 12, 20, 256)])
def test_batchnorm_train_forward_backward(lib, B, H, W, Cc):
    """Train-mode BatchNorm2d (batch statistics, biased variance) on the padded-NHWC bf16 maps and its backward against
    torch autograd; plus the ReLU mask."""
    dt = torch.bfloat16
    x = rnd((B, Cc, H, W), dt, 80, 1.5) + 0.25
    dy = rnd((B, Cc, H, W), dt, 81)
    gamma = (1.0 + 0.1 * rnd((Cc,), torch.float32, 82)).contiguous()
    beta = (0.1 * rnd((Cc,), torch.float32, 83)).contiguous()
    xr, gr, br = x.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    yr.backward(dy.float())
    xp, dyp = _pad_nhwc(x, dt), _pad_nhwc(dy, dt)
    yp, dxp = torch.zeros_like(xp), torch.zeros_like(xp)
    stats = torch.empty(2 * Cc, dtype=torch.float32).cuda()
    bst = torch.empty(2 * Cc, dtype=torch.float32).cuda()
    _lib.check(lib.lseg_op_bn_train_forward(P(xp), P(yp), P(stats), P(gamma), P(beta), B, H, W, Cc, 1e-5, stream()))
    _lib.check(lib.lseg_op_bn_train_backward(P(dyp), P(xp), P(stats), P(gamma), P(dxp), P(bst), B, H, W, Cc, 1e-5, stream()))
    torch.cuda.synchronize()
    n = B * H * W
    assert (stats[:Cc] / n - x.float().mean(dim=(0, 2, 3))).abs().max().item() <= 1e-4
    got_y = yp[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
    assert (got_y - yr.detach()).abs().max().item() <= 2e-2 * max(1.0, yr.abs().max().item())
    got_dx = dxp[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
    assert (got_dx - xr.grad).abs().max().item() <= 2e-2 * max(1.0, xr.grad.abs().max().item())
    assert (bst[:Cc] - br.grad).abs().max().item() <= 2e-3 * max(1.0, br.grad.abs().max().item())
    assert (bst[Cc:] - gr.grad).abs().max().item() <= 2e-3 * max(1.0, gr.grad.abs().max().item())
    assert yp[:, 0].abs().max().item() == 0 and dxp[:, :, -1].abs().max().item() == 0          # borders stay zero
    m = torch.empty_like(dyp)
    _lib.check(lib.lseg_op_relu_backward(P(dyp), P(xp), P(m), xp.numel(), stream()))
    torch.cuda.synchronize()
    assert torch.equal(m, torch.where(xp.float() > 0, dyp, torch.zeros_like(dyp)))


@pytest.mark.parametrize("with_gram", [True, False])
@pytest.mark.parametrize("B,K,H,W", [(2, 150, 120, 120), (1, 5, 24, 24), (3, 37, 30, 17), (1, 157, 5, 3), (2, 80, 9, 31), (1, 33, 4, 15)])
def test_corr_planes_kernel(lib, B, K, H, W, with_gram):
    """csrc/corr.hip (lseg_op_corr_planes): the pixel x text correlation on the commuted schedule -- label planes R[b, k, p] = t_k . g_p
    on the padded quarter-resolution map and the five 2x2-cell dot products of g, one pass over g, T resident in LDS.  Against fp64
    torch on the same fp16 operands (what is left: fp32 accumulation order inside the MFMAs).  Shapes: BASELINE configs[1]'s map
    (120 x 120, K = 150), every label-block instantiation (K <= 32 / <= 80 / <= 157), ragged tile edges in both directions, the largest
    K the LDS takes; interior pixels of R and every gram record must be written, nothing outside the buffers touched (guard words)."""
    Cc = 512
    HP, WP = H + 2, W + 2
    g = rnd((B, HP, WP, Cc), torch.float16, 11 + K, 0.5)
    T = rnd((K, Cc), torch.float16, 12 + K, 1.0)
    T = (T.float() / T.float().norm(dim=-1, keepdim=True)).half()
    guard = 64
    Rbuf = torch.full((B * K * HP * WP + 2 * guard,), float("nan"), dtype=torch.float32).cuda()
    Gbuf = torch.full((B * H * W * 5 + 2 * guard,), float("nan"), dtype=torch.float32).cuda()
    R = Rbuf[guard:guard + B * K * HP * WP].view(B, K, HP, WP)
    G = Gbuf[guard:guard + B * H * W * 5].view(B, H, W, 5)
    _lib.check(lib.lseg_op_corr_planes(P(g), P(T), C.c_void_p(R.data_ptr()), C.c_void_p(G.data_ptr()) if with_gram else None,
                                       B, K, H, W, Cc, stream()))
    torch.cuda.synchronize()
    assert torch.isnan(Rbuf[:guard]).all() and torch.isnan(Rbuf[-guard:]).all() and torch.isnan(Gbuf[:guard]).all() and torch.isnan(Gbuf[-guard:]).all()
    gd = g.double()
    ref = torch.einsum("kc,byxc->bkyx", T.double(), gd)
    got = R[:, :, 1:H + 1, 1:W + 1]
    assert torch.isfinite(got).all(), "an interior pixel of R was not written"
    err = (got.double() - ref[:, :, 1:H + 1, 1:W + 1]).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    border = torch.ones((HP, WP), dtype=torch.bool)
    border[1:H + 1, 1:W + 1] = False
    assert torch.isnan(R[:, :, border.cuda()]).all()                 # the border is never read by the upsample and never written here
    if with_gram:
        a = gd[:, 1:H + 1, 1:W + 1]                                  # q = (y, x)
        r_ = gd[:, 1:H + 1, 2:W + 2]                                 # (y, x+1)
        d_ = gd[:, 2:H + 2, 1:W + 1]                                 # (y+1, x)
        dr = gd[:, 2:H + 2, 2:W + 2]                                 # (y+1, x+1)
        gref = torch.stack([(a * a).sum(-1), (a * r_).sum(-1), (a * d_).sum(-1), (a * dr).sum(-1), (r_ * d_).sum(-1)], dim=-1)
        assert torch.isfinite(G).all(), "a gram record was not written"
        gerr = (G.double() - gref).abs().max().item()
        assert gerr <= 2e-5 * max(1.0, gref.abs().max().item()), gerr
    else:
        assert torch.isnan(Gbuf).all()


def test_corr_planes_rejects_what_the_lds_cannot_hold(lib):
    g = rnd((1, 6, 6, 512), torch.float16, 1)
    T = rnd((200, 512), torch.float16, 2)
    R = torch.zeros((1, 200, 6, 6)).cuda()
    assert lib.lseg_op_corr_planes(P(g), P(T), P(R), None, 1, 200, 4, 4, 512, stream()) == -5          # LSEG_ERR_UNSUPPORTED: K x 1040 B > 160 KB
    assert lib.lseg_op_corr_planes(P(g), P(T), P(R), None, 1, 8, 4, 4, 768, stream()) == -5            # other widths: the generic GEMM
