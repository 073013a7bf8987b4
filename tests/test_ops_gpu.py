"""Op-level parity tests: every HIP kernel family, called through the C ABI (foley_op_*),
against a plain PyTorch fp32 CPU statement of the same op.  fp32 operands must agree to fp32
round-off (the MFMA fp32 path is an exact FMA chain); bf16 operands are compared against the
same math on bf16-rounded inputs with a bf16-appropriate tolerance.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from foley_amd.host import packers, runtime as rt, tables
from oracle import foley_oracle as O

pytestmark = pytest.mark.gpu

F32_TOL = 2e-6
BF16_TOL = 6e-3
F16_TOL = 8e-4      # fp16 operands (precision=fp16): three more mantissa bits than bf16


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=_g(seed)) * scale


def _tol(dtype):
    return {torch.float32: F32_TOL, torch.bfloat16: BF16_TOL, torch.float16: F16_TOL}[dtype]


def _q(t, dtype):   # round a CPU reference operand through the compute dtype
    return t.to(dtype).to(torch.float32)


# ----------------------------------------------------------------------------- GEMM: plain linear
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,tile", [
    (500, 1536, 1536, 0), (500, 1536, 1536, 1), (500, 1536, 1536, 2), (500, 1536, 1536, 3),
    (37, 200, 256, 0), (130, 128, 128, 0), (10, 13824, 1536, 0), (4000, 64, 128, 4), (77, 3072, 768, 0),
    (500, 1536, 1536, 5), (500, 1536, 1536, 6), (500, 1536, 1536, 8), (37, 200, 256, 6), (130, 192, 128, 5),
    (700, 640, 4608, 5),
    # wave-specialised mainloop (bf16): 128x128 / 256x128 tiles, ragged M and N edges, K = one slice .. many
    (500, 1536, 1536, 15), (500, 1536, 1536, 19), (37, 200, 64, 15), (700, 640, 4608, 19), (1000, 136, 320, 15),
    (257, 129, 128, 19),
    # four consumer waves (64x64 / 128x64 wave tiles) + loader waves helping in the epilogue
    (500, 1536, 1536, 25), (500, 1536, 1536, 29), (37, 200, 64, 25), (700, 640, 4608, 29), (1000, 136, 320, 25), (257, 129, 128, 29),
    # 256x256 tiles on the BK = 32 mainloop (gemm_wide_impl.h, tile 32): ragged M / N edges, half-empty column passes (N = 136),
    # K = one 64-wide slice pair .. many
    (500, 1536, 1536, 32), (37, 200, 64, 32), (700, 640, 4608, 32), (1000, 136, 320, 32), (257, 136, 128, 32), (4000, 512, 192, 32),
])
def test_gemm_linear(dev, dtype, M, N, K, tile):
    if tile in (15, 19, 25, 29, 32) and dtype == torch.float32:
        pytest.skip("wave-specialised tiles are bf16 only")
    A, W, b = _rand((M, K), 1), _rand((N, K), 2, 1 / math.sqrt(K)), _rand((N,), 3, 0.1)
    ref = F.linear(_q(A, dtype), _q(W, dtype), b)
    out = torch.full((M, N), float("nan"), device=dev)
    rt.op_gemm(A.to(dev, dtype), W.to(dev, dtype), b.to(dev), out0=out, tile=tile)
    assert rel_err(out, ref) < _tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 15, 25])
@pytest.mark.parametrize("segs,segV,segS", [(10, 8, 16), (100, 8, 112), (3, 40, 41)])
def test_gemm_virtual_rows_skip_source_rows(dev, dtype, tile, segs, segV, segS):
    """Plain linear layer over VIRTUAL rows (kernels.h): row r of the product reads source row (r // segV) * segS + r % segV -
    the single-block modulation GEMM runs on the first P tokens of every (iteration, CFG half) segment of Ls tokens
    (foley_rt.hip foley_prepare step 6 / run_forward).  Every mainloop family must honour the mapping (the register-staged
    tiles once ignored it for taps == 1: invisible while both CFG halves carried identical rows)."""
    if tile in (15, 25) and dtype == torch.float32:
        pytest.skip("wave-specialised tiles are 16-bit only")
    K, N = 256, 384
    A, W, b = _rand((segs * segS, K), 71), _rand((N, K), 72, 1 / math.sqrt(K)), _rand((N,), 73, 0.1)
    idx = (torch.arange(segs)[:, None] * segS + torch.arange(segV)[None]).reshape(-1)
    ref = F.linear(_q(A, dtype)[idx], _q(W, dtype), b)
    out = torch.full((segs * segV, N), float("nan"), device=dev)
    rt.op_gemm(A.to(dev, dtype), W.to(dev, dtype), b.to(dev), out0=out, M=segs * segV, vrows=(segV, segS), tile=tile)
    assert rel_err(out, ref) < _tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [0, 3, 5, 6])
def test_gemm_transpose_detecting(dev, dtype, tile):
    """A = I with an asymmetric W: catches a swapped C/D row/column mapping (and, for the
    direct-to-LDS loop, any mismatch between the source swizzle and the fragment-read swizzle)."""
    K = 128
    A = torch.eye(K)
    W = torch.arange(192 * K, dtype=torch.float32).view(192, K) % 251 / 16.0
    out = torch.empty(K, 192, device=dev)
    rt.op_gemm(A.to(dev, dtype), W.to(dev, dtype), None, out0=out, tile=tile)
    assert torch.equal(out.cpu(), W.t().contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [0, 15, 19, 25, 29, 32])
@pytest.mark.parametrize("epi", ["store_t", "silu", "gelu", "silugate", "gate_res_vec", "gate_res_tok", "addend"])
def test_gemm_epilogues(dev, dtype, epi, tile):
    if tile and dtype == torch.float32:
        pytest.skip("wave-specialised tiles are bf16 only")
    if tile == 32 and epi in ("store_t", "silu"):
        pytest.skip("the 256x256 tiles carry the epilogues of the large-grid GEMMs only")
    M, N, K = 300, 512, 256
    clips, L = 3, 50                     # rows ordered [cfg=2][clip=3][l=50]
    A, W, b = _rand((M, K), 4), _rand((N, K), 5, 1 / math.sqrt(K)), _rand((N,), 6, 0.1)
    y = F.linear(_q(A, dtype), _q(W, dtype), b)
    Ad, Wd, bd = A.to(dev, dtype), W.to(dev, dtype), b.to(dev)
    if epi in ("store_t", "silu", "gelu"):
        out = torch.empty(M, N, device=dev, dtype=dtype)
        code = {"store_t": rt.EPI_STORE_T, "silu": rt.EPI_SILU_T, "gelu": rt.EPI_GELU_T}[epi]
        rt.op_gemm(Ad, Wd, bd, out0=out, epilogue=code, tile=tile)
        ref = {"store_t": y, "silu": F.silu(y), "gelu": F.gelu(y, approximate="tanh")}[epi]
        assert rel_err(out.float(), ref) < _tol(dtype)
    elif epi == "silugate":
        w1, w3 = W[: N // 2], W[N // 2:]
        Wp = packers.interleave_gate(w1, w3)
        out = torch.empty(M, N // 2, device=dev, dtype=dtype)
        rt.op_gemm(Ad, Wp.to(dev, dtype), None, out0=out, epilogue=rt.EPI_SILUGATE_T, tile=tile)
        ref = F.silu(F.linear(_q(A, dtype), _q(w1, dtype))) * F.linear(_q(A, dtype), _q(w3, dtype))
        assert rel_err(out.float(), ref) < _tol(dtype) * (1 if dtype == torch.float32 else 1.5)
    elif epi == "gate_res_vec":
        x0, gate = _rand((M, N), 7), _rand((N,), 8)
        x = x0.to(dev).clone()
        rt.op_gemm(Ad, Wd, bd, out0=x, epilogue=rt.EPI_GATE_RES, rb=rt.rowbcast(gate.to(dev), 0), tile=tile)
        assert rel_err(x, x0 + y * gate) < _tol(dtype)
    elif epi == "gate_res_tok":
        x0, gate = _rand((M, N), 7), _rand((2, L, N), 8)
        x = x0.to(dev).clone()
        rt.op_gemm(Ad, Wd, bd, out0=x, epilogue=rt.EPI_GATE_RES,
                   rb=rt.rowbcast(gate.to(dev), 1, rows_per_cfg=clips * L, L=L), tile=tile)
        ref = x0 + y * gate[:, None].expand(2, clips, L, N).reshape(M, N)
        assert rel_err(x, ref) < _tol(dtype)
    else:
        add = _rand((2, L, N), 9)
        out = torch.empty(M, N, device=dev)
        rt.op_gemm(Ad, Wd, bd, out0=out, rb=rt.rowbcast(add.to(dev), 1, rows_per_cfg=clips * L, L=L), tile=tile)
        ref = y + add[:, None].expand(2, clips, L, N).reshape(M, N)
        assert rel_err(out, ref) < _tol(dtype)


@pytest.mark.parametrize("fmt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("tile", [15, 19, 21, 23, 31, 32])
@pytest.mark.parametrize("case", ["linear", "ragged", "conv3", "gelu", "silugate", "gate_res_split", "qkv_split"])
@pytest.mark.parametrize("act", [torch.bfloat16, torch.float16])
def test_gemm_fp8_weight_storage(dev, fmt, tile, case, act):
    """In-kernel fp8 weight storage (reference FP8WeightWrapper, utils.py:316-366: weight kept in fp8, plain
    cast, `w.to(x.dtype)` per call): the loaders move 64-byte fp8 K-slices, the consumers widen to bf16 in
    registers.  Widening is exact and both kernels use the same K order, so the result must be BIT-IDENTICAL
    to the same GEMM on the weights widened to bf16 at load time - and therefore as close to the fp32
    statement on fp8-rounded weights as the bf16 path is."""
    if tile in (21, 23, 31) and case not in ("conv3", "gate_res_split"):
        pytest.skip("tiles 21 / 23 / 31 are the tap-fused conv k=3 kernels")
    if tile == 32 and case in ("conv3", "gate_res_split"):
        pytest.skip("tile 32 is the plain 256x256 tile (32-byte fp8 weight rows): linear layers (round 6: with the head split)")
    M, N, K = {"linear": (500, 1536, 1536), "ragged": (257, 1408, 320), "conv3": (500, 512, 3 * 256), "gelu": (300, 512, 256),
               "silugate": (300, 512, 256), "gate_res_split": (500, 1536, 3 * 512), "qkv_split": (500, 3 * 2 * 128, 256)}[case]
    conv = (250, K // 3, 3, 1) if case in ("conv3", "gate_res_split") else None
    A = _rand((M, K // 3 if conv else K), 21).to(dev, act)
    W8 = _rand((N, K), 22, 1 / math.sqrt(K)).to(fmt)
    Wq = W8.to(act)                       # exact: every fp8 value is a bf16 value and an fp16 value
    assert torch.equal(Wq.float(), W8.float())
    b = _rand((N,), 23, 0.1).to(dev)
    W8d, Wqd = W8.to(dev), Wq.to(dev)
    kw = dict(conv=conv) if conv else {}

    def run(W):
        if case in ("linear", "ragged", "conv3"):
            out = torch.full((M, N), float("nan"), device=dev)
            rt.op_gemm(A, W, b, out0=out, tile=tile, **kw)
            return out
        if case == "gelu":
            out = torch.empty(M, N, device=dev, dtype=act)
            rt.op_gemm(A, W, b, out0=out, epilogue=rt.EPI_GELU_T, tile=tile)
            return out
        if case == "silugate":
            out = torch.empty(M, N // 2, device=dev, dtype=act)
            rt.op_gemm(A, W, None, out0=out, epilogue=rt.EPI_SILUGATE_T, tile=tile)
            return out
        if case == "gate_res_split":
            x = _rand((M, N), 24).to(dev)
            slabs = torch.zeros(8, M, N, device=dev)
            gate = _rand((N,), 25).to(dev)
            ks = rt.op_gemm(A, W, b, out0=x, epilogue=rt.EPI_GATE_RES, rb=rt.rowbcast(gate, 0), tile=tile, ksplit=0,
                            partials=slabs, **kw)
            return torch.cat([x, slabs[:ks].sum(0)])
        H, L = 2, 250
        dq, dk = (torch.zeros(M // L, H, L, 128, device=dev, dtype=act) for _ in range(2))
        dvt = torch.zeros(M // L, H, 128, 256, device=dev, dtype=act)
        cos, sin = (t.to(dev) for t in tables.rope_table(L + 1))
        gq = (1 + 0.1 * _rand((128,), 26)).to(dev)
        pos = torch.arange(L, dtype=torch.int32, device=dev)
        desc = rt.qkv_split_desc(L, H, [gq, gq, None], [pos, pos, None], [dq, dk, dvt], L, 0, 1e-6, cos, sin, vt_pitch=256)
        rt.op_gemm(A, W, b, epilogue=rt.EPI_QKV_SPLIT, qkv=desc, tile=tile)
        return torch.cat([dq.flatten(), dk.flatten(), dvt.flatten()])

    y8, yq = run(W8d), run(Wqd)
    assert torch.isfinite(y8.float()).all() and torch.equal(y8, yq)
    if case in ("linear", "ragged"):
        assert rel_err(y8, F.linear(A.float().cpu(), W8.float(), b.cpu())) < _tol(act)


def test_gemm_fp8_rejects_unsupported(dev):
    """fp8 weights exist only in the wave-specialised bf16 mainloop: other tiles / fp32 operands fail loudly."""
    A = _rand((64, 128), 1).to(dev, torch.bfloat16)
    W8 = _rand((128, 128), 2).to(torch.float8_e4m3fn).to(dev)
    out = torch.empty(64, 128, device=dev)
    with pytest.raises(rt.FoleyRuntimeError):
        rt.op_gemm(A, W8, None, out0=out, tile=3)
    rt.op_gemm(A, W8, None, out0=out, tile=0)        # auto -> a wave-specialised tile
    assert rel_err(out, A.float() @ W8.float().t()) < BF16_TOL


@pytest.mark.parametrize("ksplit", [0, 1, 2, 5])
@pytest.mark.parametrize("conv", [False, True, 11, 13])
def test_gemm_split_k(dev, ksplit, conv):
    """Gated-residual epilogue with K ranges accumulated by fp32 atomics (bf16 mode), incl. the
    conv addressing whose tap cursor must start mid-way for the later ranges."""
    B, L, C, N = 2, 50, 256, 192
    x, w, b = _rand((B, L, C), 24), _rand((N, C, 3), 25, 1 / math.sqrt(3 * C)), _rand((N,), 26, 0.1)
    dt = torch.bfloat16
    res0, gate = _rand((B * L, N), 27), _rand((N,), 28)
    if conv:
        y = O.conv1d_cl(_q(x, dt), _q(w, dt), b, 1).reshape(B * L, N)
        Wp, kw = packers.conv_to_gemm(w), dict(conv=(L, C, 3, 1))
        if conv in (11, 13):
            kw["tile"] = conv    # tap-fused conv kernel
    else:
        y = F.linear(_q(x, dt).reshape(B * L, C), _q(w[:, :, 0], dt), b)
        Wp, kw = w[:, :, 0].contiguous(), {}
    out = res0.to(dev).clone()
    rt.op_gemm(x.reshape(B * L, C).to(dev, dt), Wp.to(dev, dt), b.to(dev), out0=out, epilogue=rt.EPI_GATE_RES,
               rb=rt.rowbcast(gate.to(dev), 0), ksplit=ksplit, **kw)
    assert rel_err(out, res0 + y * gate) < 1e-5


@pytest.mark.parametrize("ksplit", [0, 2, 3, 7])
@pytest.mark.parametrize("conv", [False, True, 11, 13, 15, 19, 21, 22, 23, 24, 31, -32])
@pytest.mark.parametrize("tok_gate", [False, True])
@pytest.mark.parametrize("slab_dt", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_deferred_split_k(dev, ksplit, conv, tok_gate, slab_dt):
    """Deferred split-K: the gated-residual GEMM leaves raw partial products per K range, the next
    LayerNorm applies x += gate * (sum + bias) in place before normalising (no atomics).  The slabs are fp32, or - the
    sampler's default in the 16-bit modes - stored in the operand type (half the slab traffic): k rounded partials."""
    ncfg, clips, L, C, N = 2, 1, 50, 256, 192
    B = ncfg * clips
    x, w, b = _rand((B, L, C), 24), _rand((N, C, 3), 25, 1 / math.sqrt(3 * C)), _rand((N,), 26, 0.1)
    dt = torch.float16 if slab_dt == torch.float16 else torch.bfloat16
    h16 = slab_dt != torch.float32
    tol_s = {torch.float32: 1e-5, torch.bfloat16: 4e-3, torch.float16: 5e-4}[slab_dt]
    res0 = _rand((B * L, N), 27)
    if tok_gate:   # per-(cfg, token) gate rows, as the single-stream blocks use
        gate = _rand((ncfg * L, N), 28)
        g_full = gate.view(ncfg, 1, L, N).expand(ncfg, clips, L, N).reshape(B * L, N)
        rb = rt.rowbcast(gate.to(dev), 1, rows_per_cfg=clips * L, L=L)
    else:
        gate = _rand((N,), 28)
        g_full = gate
        rb = rt.rowbcast(gate.to(dev), 0)
    if conv and conv > 0:
        y = O.conv1d_cl(_q(x, dt), _q(w, dt), b, 1).reshape(B * L, N)
        Wp, kw = packers.conv_to_gemm(w), dict(conv=(L, C, 3, 1))
        if conv in (11, 13, 15, 19, 21, 22, 23, 24, 31):
            kw["tile"] = conv    # tap-fused / wave-specialised conv addressing
    else:
        y = F.linear(_q(x, dt).reshape(B * L, C), _q(w[:, :, 0], dt), b)
        Wp, kw = w[:, :, 0].contiguous(), ({"tile": -conv} if conv else {})     # negative: a plain linear layer on that tile
    xres = res0.to(dev).clone()
    slabs = torch.full((8, B * L, N), float("nan"), device=dev, dtype=slab_dt)
    used = rt.op_gemm(x.reshape(B * L, C).to(dev, dt), Wp.to(dev, dt), b.to(dev), out0=xres, epilogue=rt.EPI_GATE_RES,
                      rb=rb, ksplit=ksplit, partials=slabs, **kw)
    assert 1 <= used <= 8 and (ksplit == 0 or used == ksplit)
    want_x = res0 + y * g_full
    shift, scale = _rand((N,), 29), _rand((N,), 30)
    out = torch.empty(B * L, N, device=dev, dtype=slab_dt)      # 16-bit slabs carry the LayerNorm's output type
    if used > 1:
        assert torch.equal(xres.cpu(), res0)                       # residual untouched by the GEMM
        assert rel_err(slabs[:used].float().sum(0), y - b) < tol_s # raw products, bias not included
        rt.op_ln_mod_pending(xres, 1e-6, rt.rowbcast(shift.to(dev), 0), rt.rowbcast(scale.to(dev), 0), out, slabs, used,
                             b.to(dev), rb)
    else:
        rt.op_ln_mod(xres, 1e-6, rt.rowbcast(shift.to(dev), 0), rt.rowbcast(scale.to(dev), 0), out)
    assert rel_err(xres, want_x) < (tol_s if used > 1 else 1e-5)
    ref = F.layer_norm(want_x, (N,), eps=1e-6) * (1 + scale) + shift
    assert rel_err(out.float(), ref) < (1e-4 if not h16 else 1.5 * tol_s + 4e-3 * (slab_dt == torch.bfloat16))


# ----------------------------------------------------------------------------- GEMM: conv addressing
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [0, 3, 5, 6, 11, 13, 15, 19, 21, 22, 23, 24, 31])
@pytest.mark.parametrize("B,L,Cin,Cout", [(2, 50, 256, 384), (3, 7, 128, 64), (1, 250, 1536, 256), (5, 33, 128, 200), (2, 250, 64, 128),
                                          (4, 129, 192, 320), (16, 250, 128, 256)])
def test_conv3_channels_last(dev, dtype, tile, B, L, Cin, Cout):
    if tile in (15, 19, 21, 22, 23, 24, 31) and dtype == torch.float32:
        pytest.skip("wave-specialised tiles are bf16 only")
    """ChannelLastConv1d k=3 pad=1 (mlp_layers.py:104-110) as a GEMM over overlapping rows
    (register-staged and direct-to-LDS mainloops)."""
    x, w, b = _rand((B, L, Cin), 10), _rand((Cout, Cin, 3), 11, 1 / math.sqrt(3 * Cin)), _rand((Cout,), 12, 0.1)
    ref = O.conv1d_cl(_q(x, dtype), _q(w, dtype), b, 1).reshape(B * L, Cout)
    out = torch.empty(B * L, Cout, device=dev)
    rt.op_gemm(x.reshape(B * L, Cin).to(dev, dtype), packers.conv_to_gemm(w).to(dev, dtype), b.to(dev),
               out0=out, conv=(L, Cin, 3, 1), tile=tile)
    assert rel_err(out, ref) < _tol(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [0, 21, 22, 23, 31])
@pytest.mark.parametrize("B,L,Cin,Cout", [(2, 250, 1536, 512), (3, 70, 128, 256), (4, 129, 192, 384)])
def test_conv3_silu_gate(dev, dtype, tile, B, L, Cin, Cout):
    """ConvMLP w1 / w3 (mlp_layers.py:113-149): silu(conv3(x, w1)) * conv3(x, w3) in one launch over the
    interleaved weight, on every tap-fused tile form (128x128, 256x64, 256x128)."""
    x = _rand((B, L, Cin), 60)
    w1, w3 = (_rand((Cout, Cin, 3), 61 + i, 1 / math.sqrt(3 * Cin)) for i in range(2))
    c = lambda w: O.conv1d_cl(_q(x, dtype), _q(w, dtype), None, 1).reshape(B * L, Cout)
    ref = F.silu(c(w1)) * c(w3)
    Wp = packers.interleave_gate(packers.conv_to_gemm(w1), packers.conv_to_gemm(w3))
    out = torch.full((B * L, Cout), float("nan"), device=dev, dtype=dtype)
    rt.op_gemm(x.reshape(B * L, Cin).to(dev, dtype), Wp.to(dev, dtype), None, out0=out, conv=(L, Cin, 3, 1),
               epilogue=rt.EPI_SILUGATE_T, tile=tile)
    assert rel_err(out.float(), ref) < _tol(dtype) * 1.5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("stride,B,Tin,Cin,Cout", [(2, 2, 60, 64, 128), (3, 1, 51, 32, 64), (5, 2, 40, 128, 96), (8, 1, 64, 64, 128)])
def test_strided_conv_as_gemm(dev, dtype, stride, B, Tin, Cin, Cout):
    """DAC EncoderBlock down-sampling conv (dac.py:55-61): k = 2s, stride s, pad ceil(s/2), as a GEMM
    whose virtual rows advance s source rows (zero padding through the operand range check)."""
    if dtype != torch.float32 and Cin % 64:
        pytest.skip("16-bit K-slices are 64 channels wide")
    x, w, b = _rand((B, Cin, Tin), 40), _rand((Cout, Cin, 2 * stride), 41, 1 / math.sqrt(2 * stride * Cin)), _rand((Cout,), 42, 0.1)
    ref = F.conv1d(_q(x, dtype), _q(w, dtype), b, stride=stride, padding=math.ceil(stride / 2)).transpose(1, 2)
    Tout = Tin // stride
    assert ref.shape[1] == Tout
    xs = x.transpose(1, 2).contiguous().to(dev, dtype)                   # [B, Tin, Cin] time-major
    out = torch.full((B * Tout, Cout), float("nan"), device=dev)
    rt.op_gemm(xs, packers.conv_to_gemm(w).to(dev, dtype), b.to(dev), out0=out, sconv=(Tin, Cin, stride))
    assert rel_err(out.view(B, Tout, Cout), ref) < _tol(dtype)


@pytest.mark.parametrize("dil", [1, 3, 9])
@pytest.mark.parametrize("B,T,C", [(2, 100, 64), (1, 37, 128)])
def test_dac_conv7_snake(dev, dil, B, T, C):
    """Dilated conv k=7 + snake epilogue; residual 1x1 conv + snake (dac.py:28-44)."""
    x, w, b = _rand((B, C, T), 13), _rand((C, C, 7), 14, 1 / math.sqrt(7 * C)), _rand((C,), 15, 0.1)
    a2 = 1 + 0.2 * _rand((C,), 16)
    ref = O.snake(F.conv1d(x, w, b, dilation=dil, padding=3 * dil), a2.view(1, C, 1)).transpose(1, 2)
    xs = x.transpose(1, 2).contiguous().to(dev)                       # [B, T, C]
    out1 = torch.empty(B, T, C, device=dev)
    rt.op_gemm(xs, packers.conv_to_gemm(w).to(dev), b.to(dev), out1=out1, conv=(T, C, 7, dil),
               epilogue=rt.EPI_DAC, alpha=a2.to(dev), alphaC=C)
    assert rel_err(out1, ref) < 3e-6
    # 1x1 conv with residual, both outputs
    w1, b1 = _rand((C, C, 1), 17, 1 / math.sqrt(C)), _rand((C,), 18, 0.1)
    res = _rand((B, T, C), 19)
    y = F.conv1d(x, w1, b1).transpose(1, 2) + res
    xo = res.to(dev).clone()
    so = torch.empty(B, T, C, device=dev)
    rt.op_gemm(xs.view(B * T, C), w1.squeeze(-1).contiguous().to(dev), b1.to(dev), out0=xo, out1=so, res=xo,
               epilogue=rt.EPI_DAC, alpha=a2.to(dev), alphaC=C)
    assert rel_err(xo, y) < 3e-6
    assert rel_err(so, O.snake(y.transpose(1, 2), a2.view(1, C, 1)).transpose(1, 2)) < 3e-6


@pytest.mark.parametrize("s", [2, 3, 4, 5, 8])
def test_dac_conv_transpose(dev, s):
    """ConvTranspose1d(k=2s, stride s, pad ceil(s/2), out_pad s%2) (dac.py:102-109) as one GEMM."""
    B, Tin, Cin, Cout = 2, 23, 128, 64
    x, w, b = _rand((B, Cin, Tin), 20), _rand((Cin, Cout, 2 * s), 21, 1 / math.sqrt(2 * Cin)), _rand((Cout,), 22, 0.1)
    a = 1 + 0.2 * _rand((Cout,), 23)
    y = F.conv_transpose1d(x, w, b, stride=s, padding=math.ceil(s / 2), output_padding=s % 2)
    assert y.shape[-1] == Tin * s
    X = torch.full((B, Tin * s, Cout), float("nan"), device=dev)
    S = torch.full((B, Tin * s, Cout), float("nan"), device=dev)
    rt.op_gemm(x.transpose(1, 2).contiguous().to(dev), packers.convT_to_gemm(w, s).to(dev), b.repeat(s).to(dev),
               out0=X, out1=S, convT=(Tin, Cin, s, Cout), epilogue=rt.EPI_DAC, alpha=a.to(dev), alphaC=Cout)
    assert rel_err(X, y.transpose(1, 2)) < 3e-6
    assert rel_err(S, O.snake(y, a.view(1, Cout, 1)).transpose(1, 2)) < 3e-6


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,Sq,Skv,split,kv_bdiv", [
    (2, 2, 58, 58, 8, 1), (4, 3, 290, 77, 40, 2), (2, 12, 250, 250, 0, 1), (1, 1, 33, 31, 5, 1)])
def test_attention(dev, out_dtype, B, H, Sq, Skv, split, kv_bdiv):
    q, k, v = _rand((B, H, Sq, 128), 30), _rand((B // kv_bdiv, H, Skv, 128), 31), _rand((B // kv_bdiv, H, Skv, 128), 32)
    ke, ve = k.repeat_interleave(kv_bdiv, 0), v.repeat_interleave(kv_bdiv, 0)
    ref = O.sdpa(q, ke, ve).transpose(1, 2).reshape(B, Sq, H * 128)
    oa = torch.full((B, max(split, 1), H * 128), float("nan"), device=dev, dtype=out_dtype)
    ob = torch.full((B, Sq - split, H * 128), float("nan"), device=dev, dtype=out_dtype)
    rt.op_attention(q.to(dev), k.to(dev), v.to(dev), oa, ob, split, kv_bdiv)
    tol = {torch.float32: 3e-6, torch.bfloat16: 4e-3, torch.float16: 5e-4}[out_dtype]
    if split:
        assert rel_err(oa.float(), ref[:, :split]) < tol
    assert rel_err(ob.float(), ref[:, split:]) < tol


@pytest.mark.parametrize("B,H,Sq,Skv,split,kv_bdiv", [
    (2, 2, 58, 58, 8, 1), (4, 3, 290, 77, 40, 2), (2, 12, 250, 250, 0, 1), (1, 1, 33, 31, 5, 1), (1, 2, 100, 1500, 0, 1),
    # >= 256 workgroups of 128 queries: the wide kernel (K / V^T tiles staged in LDS, no key split)
    (16, 12, 290, 290, 40, 1), (16, 12, 290, 77, 40, 8), (32, 4, 257, 95, 3, 1), (64, 2, 129, 33, 0, 2),
    # config C5 (30 s, CFG pair): the wide kernel at its real key-tile counts - 55 / 47 tiles of online-softmax
    # rescaling with a ragged last tile (1740 = 54*32 + 12, 1500 = 46*32 + 28), and the 77 cached text keys
    (2, 12, 1740, 1740, 240, 1), (2, 12, 1500, 1500, 0, 1), (2, 12, 1740, 77, 240, 1),
    # the 64-keys-per-iteration form of long sequences on small grids (attn_bf16_long_kernel: the two C5 self-attention shapes above
    # and these): last tile with 33 / 24 / 64 valid keys, odd tile counts, a shared K / V batch
    (3, 12, 1100, 545, 0, 1), (2, 12, 1400, 600, 100, 1), (4, 6, 1300, 1024, 0, 2)])
@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
def test_attention_bf16(dev, B, H, Sq, Skv, split, kv_bdiv, half):
    """Throughput kernels: bf16 Q/K, transposed bf16 V with a padded pitch; 4-wave key split for
    small grids, 128-query workgroups with LDS-staged key tiles for large ones."""
    q, k, v = _rand((B, H, Sq, 128), 30), _rand((B // kv_bdiv, H, Skv, 128), 31), _rand((B // kv_bdiv, H, Skv, 128), 32)
    qb, kb, vb = (t.to(half) for t in (q, k, v))
    ke, ve = kb.float().repeat_interleave(kv_bdiv, 0), vb.float().repeat_interleave(kv_bdiv, 0)
    ref = O.sdpa(qb.float(), ke, ve).transpose(1, 2).reshape(B, Sq, H * 128)
    pitch = (Skv + 31) // 32 * 32
    vt = torch.full((B // kv_bdiv, H, 128, pitch), 7.0, dtype=half)       # finite garbage in the pad
    vt[..., :Skv] = vb.transpose(2, 3)
    oa = torch.full((B, max(split, 1), H * 128), float("nan"), device=dev, dtype=half)
    ob = torch.full((B, Sq - split, H * 128), float("nan"), device=dev, dtype=half)
    rt.op_attention(qb.to(dev), kb.to(dev), vt.to(dev), oa, ob, split, kv_bdiv)
    if split:
        assert rel_err(oa.float(), ref[:, :split]) < (1e-2 if half == torch.bfloat16 else 2e-3)
    assert rel_err(ob.float(), ref[:, split:]) < (1e-2 if half == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,Sq,Skv,kv_bdiv", [
    # head dim 64, the shapes of the conditioning encoders (host/encoders_hip.py): Synchformer time groups (8 queries x 9
    # keys), space groups (196 x 197), the CLS query over all 1569 tokens, the spatial aggregation layer (197 x 197),
    # SigLIP2 (1024 x 1024) and its pooling probe (1 x 1024)
    (392, 12, 8, 9, 1), (16, 12, 196, 197, 1), (2, 12, 1, 1569, 1), (16, 12, 197, 197, 1), (2, 12, 1024, 1024, 1), (3, 12, 1, 1024, 1),
    (5, 3, 37, 70, 1), (4, 2, 130, 33, 2)])
def test_attention_head_dim_64(dev, dtype, B, H, Sq, Skv, kv_bdiv):
    """foley_op_attention_hd with head_dim 64 (ViT-B encoders; reference feature_utils.py:63-108): the fp32 MFMA kernel
    and the LDS-staged 16-bit kernel templated on the head dim, against softmax(Q K^T / 8) V."""
    q, k, v = _rand((B, H, Sq, 64), 50), _rand((B // kv_bdiv, H, Skv, 64), 51), _rand((B // kv_bdiv, H, Skv, 64), 52)
    qq, kq, vq = (_q(t, dtype) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qq, kq.repeat_interleave(kv_bdiv, 0), vq.repeat_interleave(kv_bdiv, 0))
    ref = ref.transpose(1, 2).reshape(B, Sq, H * 64)
    if dtype == torch.float32:
        vd = v.to(dev)
    else:
        pitch = (Skv + 31) // 32 * 32
        vd = torch.full((B // kv_bdiv, H, 64, pitch), 3.0, dtype=dtype)
        vd[..., :Skv] = v.to(dtype).transpose(2, 3)
        vd = vd.to(dev)
    out = torch.full((B, Sq, H * 64), float("nan"), device=dev, dtype=dtype)
    rt.op_attention(q.to(dev, dtype), k.to(dev, dtype), vd, out, out, 0, kv_bdiv)
    tol = {torch.float32: 3e-6, torch.bfloat16: 1e-2, torch.float16: 2e-3}[dtype]
    assert rel_err(out.float(), ref) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [0, 15, 25])
def test_gemm_exact_gelu_epilogue(dev, dtype, tile):
    """foley_gemm_desc.gelu_erf: nn.GELU() (erf form) of the encoders' MLPs instead of the DiT's tanh form."""
    if tile and dtype == torch.float32:
        pytest.skip("wave-specialised tiles are 16-bit only")
    M, N, K = 300, 512, 256
    A, W, b = _rand((M, K), 4), _rand((N, K), 5, 1 / math.sqrt(K)), _rand((N,), 6, 0.5)
    y = F.linear(_q(A, dtype), _q(W, dtype), b)
    out = torch.empty(M, N, device=dev, dtype=dtype)
    rt.op_gemm(A.to(dev, dtype), W.to(dev, dtype), b.to(dev), out0=out, epilogue=rt.EPI_GELU_T, tile=tile, gelu_erf=True)
    assert rel_err(out.float(), F.gelu(y)) < _tol(dtype)
    assert rel_err(out.float(), F.gelu(y, approximate="tanh")) > rel_err(out.float(), F.gelu(y)) or dtype == torch.bfloat16


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_exact_gelu_fast_form_error_bound(dev, dtype):
    """common.h::gelu_erf_fast (16-bit outputs) over x in [-8, 8]: the Abramowitz-Stegun 7.1.26 erfc form has an ABSOLUTE error
    bound of 1.5e-7 on erfc, i.e. <= 0.5 |x| 1.5e-7 <= 6e-7 on GELU(x) - far below a 16-bit ulp for |GELU| > 2e-3 (bf16) / 2e-2 (fp16), but a LARGE
    relative error in the negative tail (x <= -5, |GELU| ~ 1e-6: tens of percent).  The bound that holds everywhere is absolute:
    |out - GELU(x)| <= half a 16-bit ulp of the result + 8e-7.  x is fed through a GEMM with a one-hot weight (y = x exactly)."""
    M, N, K = 4096, 64, 64
    x = torch.linspace(-8.0, 8.0, M).to(dtype)                      # representable inputs
    A = torch.zeros(M, K, dtype=dtype)
    A[:, 0] = x
    W = torch.zeros(N, K, dtype=dtype)
    W[:, 0] = 1.0
    out = torch.empty(M, N, device=dev, dtype=dtype)
    rt.op_gemm(A.to(dev), W.to(dev), torch.zeros(N, device=dev), out0=out, epilogue=rt.EPI_GELU_T, gelu_erf=True)
    xd = x.double()
    ref = 0.5 * xd * (1.0 + torch.erf(xd / math.sqrt(2.0)))
    got = out[:, 0].double().cpu()
    assert torch.equal(out.cpu(), out[:, :1].expand(M, N).cpu())    # every column saw the same y
    mant = 8 if dtype == torch.bfloat16 else 11
    ulp = torch.maximum(torch.ldexp(torch.ones_like(ref), torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - (mant - 1)),
                        torch.full_like(ref, 2.0 ** -24 if dtype == torch.float16 else 0.0))   # fp16 subnormal spacing
    err = (got - ref).abs()
    assert bool((err <= 0.5 * ulp + 8e-7).all()), float((err - 0.5 * ulp).max())
    main = ulp >= 1.2e-5      # |GELU| > ~2e-3 (bf16) / ~1.6e-2 (fp16): 6e-7 is < 5 % of an ulp there - the form IS exact to 16-bit rounding
    assert bool(main.sum() > M // 2) and bool((err[main] <= 0.56 * ulp[main]).all()), float((err[main] / ulp[main]).max())


def test_attention_bf16_spiky(dev):
    q, k, v = _rand((1, 1, 64, 128), 33), _rand((1, 1, 200, 128), 34), _rand((1, 1, 200, 128), 35)
    k[0, 0, 150] = q[0, 0, 7] * 5.0
    k[0, 0, 40] = q[0, 0, 9] * 3.0
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    ref = O.sdpa(qb.float(), kb.float(), vb.float()).transpose(1, 2).reshape(1, 64, 128)
    vt = torch.zeros(1, 1, 128, 224, dtype=torch.bfloat16)
    vt[..., :200] = vb.transpose(2, 3)
    ob = torch.empty(1, 64, 128, device=dev, dtype=torch.bfloat16)
    rt.op_attention(qb.to(dev), kb.to(dev), vt.to(dev), ob, ob, 0)
    assert rel_err(ob.float(), ref) < 1e-2


def test_attention_long_kernel_rescales(dev):
    """attn_bf16_long_kernel (64 keys per iteration): spikes placed in different 64-key tiles - and in both 32-key halves of one - force
    the running maximum of chosen queries to jump late in the key walk (the rare accumulator-rescale branch; guide rule 26)."""
    B, H, Sq, Skv = 2, 12, 1400, 640
    q, k, v = _rand((B, H, Sq, 128), 36), _rand((B, H, Skv, 128), 37), _rand((B, H, Skv, 128), 38)
    for (b, h, key, qi, amp) in ((0, 0, 70, 5, 4.0), (0, 0, 300, 5, 6.0), (0, 0, 610, 5, 9.0), (1, 7, 100, 1399, 5.0), (1, 7, 130, 1399, 8.0),
                                 (1, 3, 639, 700, 7.0)):
        k[b, h, key] = q[b, h, qi] * amp
    qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
    ref = O.sdpa(qb.float(), kb.float(), vb.float()).transpose(1, 2).reshape(B, Sq, H * 128)
    vt = vb.transpose(2, 3).contiguous()
    ob = torch.full((B, Sq, H * 128), float("nan"), device=dev, dtype=torch.bfloat16)
    rt.op_attention(qb.to(dev), kb.to(dev), vt.to(dev), ob, ob, 0)
    assert rel_err(ob.float(), ref) < 1e-2
    for (b, h, qi) in ((0, 0, 5), (1, 7, 1399), (1, 3, 700)):      # the spiked rows themselves
        assert rel_err(ob[b, qi, h * 128:(h + 1) * 128].float(), ref[b, qi, h * 128:(h + 1) * 128]) < 2e-2


def test_attention_spiky_scores(dev):
    """Forces the online-softmax running max to jump between key tiles."""
    q, k, v = _rand((1, 1, 64, 128), 33), _rand((1, 1, 200, 128), 34), _rand((1, 1, 200, 128), 35)
    k[0, 0, 150] = q[0, 0, 7] * 5.0
    k[0, 0, 40] = q[0, 0, 9] * 3.0
    ref = O.sdpa(q, k, v).transpose(1, 2).reshape(1, 64, 128)
    ob = torch.empty(1, 64, 128, device=dev)
    rt.op_attention(q.to(dev), k.to(dev), v.to(dev), ob, ob, 0)
    assert rel_err(ob, ref) < 3e-6


# ----------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D,eps", [(1536, 1e-6), (256, 1e-5), (1408, 1e-6)])
def test_ln_mod(dev, out_dtype, D, eps):
    clips, L = 2, 13
    M = 2 * clips * L
    x = _rand((M, D), 40) * 3 + 0.5
    tol = {torch.float32: 2e-6, torch.bfloat16: 4e-3, torch.float16: 5e-4}[out_dtype]
    # plain LayerNorm
    out = torch.empty(M, D, device=dev, dtype=out_dtype)
    rt.op_ln_mod(x.to(dev), eps, None, None, out)
    assert rel_err(out.float(), O.layer_norm(x, eps)) < tol
    # vector shift/scale (triple blocks)
    sh, sc = _rand((D,), 41), _rand((D,), 42) * 0.3
    rt.op_ln_mod(x.to(dev), eps, rt.rowbcast(sh.to(dev)), rt.rowbcast(sc.to(dev)), out)
    assert rel_err(out.float(), O.layer_norm(x, eps) * (1 + sc) + sh) < tol
    # per-(cfg, token) shift/scale taken from a wider table (single blocks: chunks of [.., 6D])
    tab = _rand((2, L, 6 * D), 43) * 0.3
    tb = tab.to(dev)
    rt.op_ln_mod(x.to(dev), eps, rt.rowbcast(tb[..., 3 * D:], 1, clips * L, L, ld=6 * D),
                 rt.rowbcast(tb[..., 4 * D:], 1, clips * L, L, ld=6 * D), out)
    e = lambda t: t[:, None].expand(2, clips, L, D).reshape(M, D)
    ref = O.layer_norm(x, eps) * (1 + e(tab[..., 4 * D:5 * D])) + e(tab[..., 3 * D:4 * D])
    assert rel_err(out.float(), ref) < tol


@pytest.mark.parametrize("dur", [1.0, 1.1, 2.7, 3.3, 5.0, 6.6, 13.8, 30.0])
def test_rowbcast_nearest_exact_mode(dev, dur):
    """RowBcast mode 2: the operand keeps the Ls sync-token rows and every audio frame l addresses row
    nearest_exact(l) in the kernel (float32, as F.interpolate(mode='nearest-exact'), hifi_foley.py:759-762).
    Must be bit-identical to mode 1 on the operand up-sampled on the host with tables.nearest_exact_index -
    LayerNorm shift / scale, the pending-gate path, the GEMM addend and the gated residual; the durations
    include the ones where a float64 index formula picks a different row."""
    from foley_amd.host import config as Cc
    La, _, Ls = Cc.lengths(dur)
    idx = tables.nearest_exact_index(La, Ls).long()
    assert torch.equal(idx, torch.nn.functional.interpolate(torch.arange(Ls, dtype=torch.float32).view(1, 1, Ls), size=La,
                                                            mode="nearest-exact").view(-1).long())
    clips, D = 2, 768
    M = 2 * clips * La
    x = (_rand((M, D), 60) * 2 + 0.3).to(dev)
    tab = (_rand((2, Ls, 3 * D), 61) * 0.3).to(dev)                 # [cfg, Ls, 3D]: shift | scale | gate chunks
    up = tab[:, idx.to(dev)].contiguous()                           # [cfg, La, 3D]
    rb2 = lambda c: rt.rowbcast(tab[..., c * D:], 2, clips * La, La, ld=3 * D, Ls=Ls)
    rb1 = lambda c: rt.rowbcast(up[..., c * D:], 1, clips * La, La, ld=3 * D)
    for odt in (torch.float32, torch.bfloat16):
        a, b = (torch.empty(M, D, device=dev, dtype=odt) for _ in range(2))
        rt.op_ln_mod(x, 1e-6, rb2(0), rb2(1), a)
        rt.op_ln_mod(x, 1e-6, rb1(0), rb1(1), b)
        assert torch.equal(a, b)
    # periodic token rows (empty sync features repeat every 8 tokens): only 8 rows per cfg are stored
    tab8 = tab[:, :8].contiguous()
    up8 = tab8[:, (idx % 8).to(dev)].contiguous()
    a, b = (torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(2))
    rt.op_ln_mod(x, 1e-6, rt.rowbcast(tab8[..., 0:], 2, clips * La, La, ld=3 * D, Ls=Ls, period=8),
                 rt.rowbcast(tab8[..., D:], 2, clips * La, La, ld=3 * D, Ls=Ls, period=8), a)
    rt.op_ln_mod(x, 1e-6, rt.rowbcast(up8[..., 0:], 1, clips * La, La, ld=3 * D), rt.rowbcast(up8[..., D:], 1, clips * La, La, ld=3 * D), b)
    assert torch.equal(a, b)
    # a video clip under CFG (ABI 11, periodic_cfgs = 1): the unconditional half stores its 8 periodic rows, the conditional half
    # all Ls rows right behind them - LayerNorm operands, the pending gate and both GEMM epilogue decoders
    mixed = torch.cat((tab[0, :8], tab[1]), dim=0).contiguous()      # [8 + Ls, 3D]
    upm = torch.stack((tab[0, :8][(idx % 8).to(dev)], tab[1][idx.to(dev)])).contiguous()
    rbm = lambda c: rt.rowbcast(mixed[..., c * D:], 2, clips * La, La, ld=3 * D, Ls=Ls, period=8, periodic_cfgs=1)
    rbu = lambda c: rt.rowbcast(upm[..., c * D:], 1, clips * La, La, ld=3 * D)
    a, b = (torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(2))
    rt.op_ln_mod(x, 1e-6, rbm(0), rbm(1), a)
    rt.op_ln_mod(x, 1e-6, rbu(0), rbu(1), b)
    assert torch.equal(a, b)
    slabs_m = (_rand((2, M, D), 65) * 0.1).to(dev)
    xa, xb = x.clone(), x.clone()
    rt.op_ln_mod_pending(xa, 1e-6, rbm(0), rbm(1), a, slabs_m, 2, None, rbm(2))
    rt.op_ln_mod_pending(xb, 1e-6, rbu(0), rbu(1), b, slabs_m, 2, None, rbu(2))
    assert torch.equal(a, b) and torch.equal(xa, xb)
    Am, Wm = (_rand((M, 128), 66)).to(dev), (_rand((D, 128), 67) / 128 ** 0.5).to(dev)
    for dt in (torch.float32, torch.bfloat16):
        ga, gb = x.clone(), x.clone()
        rt.op_gemm(Am.to(dt), Wm.to(dt), None, out0=ga, epilogue=rt.EPI_GATE_RES, rb=rbm(2), ksplit=1)
        rt.op_gemm(Am.to(dt), Wm.to(dt), None, out0=gb, epilogue=rt.EPI_GATE_RES, rb=rbu(2), ksplit=1)
        assert torch.equal(ga, gb)
    # pending split-K slabs + per-token gate (single blocks), then the same LayerNorm
    slabs = (_rand((2, M, D), 62) * 0.1).to(dev)
    xa, xb = x.clone(), x.clone()
    a, b = (torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(2))
    rt.op_ln_mod_pending(xa, 1e-6, rb2(0), rb2(1), a, slabs, 2, None, rb2(2))
    rt.op_ln_mod_pending(xb, 1e-6, rb1(0), rb1(1), b, slabs, 2, None, rb1(2))
    assert torch.equal(a, b) and torch.equal(xa, xb)
    # GEMM epilogues: addend (audio_embedder + add_sync) and gated residual, vector and scalar epilogue paths
    K = 128
    A, W = (_rand((M, K), 63)).to(dev), (_rand((D, K), 64) / K ** 0.5).to(dev)
    for dt in (torch.float32, torch.bfloat16):
        oa, ob = (torch.empty(M, D, device=dev) for _ in range(2))
        rt.op_gemm(A.to(dt), W.to(dt), None, out0=oa, rb=rb2(0))
        rt.op_gemm(A.to(dt), W.to(dt), None, out0=ob, rb=rb1(0))
        assert torch.equal(oa, ob)
        ga, gb = x.clone(), x.clone()
        rt.op_gemm(A.to(dt), W.to(dt), None, out0=ga, epilogue=rt.EPI_GATE_RES, rb=rb2(2), ksplit=1)
        rt.op_gemm(A.to(dt), W.to(dt), None, out0=gb, epilogue=rt.EPI_GATE_RES, rb=rb1(2), ksplit=1)
        assert torch.equal(ga, gb)


def test_rowbcast_view_pointer(dev):
    """rowbcast() must pass the *view's* data pointer (chunk offset inside a wider table)."""
    t = torch.arange(24, dtype=torch.float32, device=dev).view(2, 12)
    v = t[:, 4:8]
    assert v.data_ptr() == t.data_ptr() + 16


@pytest.mark.parametrize("eps", [1e-6, 1.1920928955078125e-07])
def test_qkv_split(dev, eps):
    """'(K H D)' split + RMSNorm + RoPE into [B, H, S, 128] at a token offset."""
    B, L, H, Lv = 2, 11, 3, 4
    S = L + Lv
    qkv = _rand((B * L, 3 * H * 128), 50)
    gq, gk = 1 + 0.1 * _rand((128,), 51), 1 + 0.1 * _rand((128,), 52)
    pos = (2 * torch.arange(L)).to(torch.int32)
    cos, sin = tables.rope_table(2 * L + 1)
    q, k, v = qkv.view(B, L, 3, H, 128).unbind(2)
    c2, s2 = cos[pos.long()].repeat_interleave(2, 1), sin[pos.long()].repeat_interleave(2, 1)
    rq = O.apply_rope(O.rms_norm(q, gq, eps), c2, s2).transpose(1, 2)       # [B, H, L, 128]
    rk = O.apply_rope(O.rms_norm(k, gk, eps), c2, s2).transpose(1, 2)
    dq, dk, dv = (torch.zeros(B, H, S, 128, device=dev) for _ in range(3))
    rt.op_qkv_split(qkv.to(dev), L, H, [gq.to(dev), gk.to(dev), None], [pos.to(dev), pos.to(dev), None],
                    [dq, dk, dv], S, Lv, eps, cos.to(dev), sin.to(dev))
    assert rel_err(dq[:, :, Lv:], rq) < 2e-6 and rel_err(dk[:, :, Lv:], rk) < 2e-6
    assert torch.equal(dv[:, :, Lv:].cpu(), v.transpose(1, 2))
    assert float(dq[:, :, :Lv].abs().max()) == 0.0


def test_qkv_split_bf16_transposed_v(dev):
    """bf16 outputs with V written transposed [B, H, 128, pitch] for the bf16 attention kernel."""
    B, L, H, Lv = 2, 11, 3, 4
    S, pitch = L + Lv, 32
    qkv = _rand((B * L, 3 * H * 128), 50)
    g = 1 + 0.1 * _rand((128,), 51)
    pos = torch.arange(L, dtype=torch.int32)
    cos, sin = tables.rope_table(L + 1)
    q, k, v = qkv.view(B, L, 3, H, 128).unbind(2)
    c2, s2 = cos[:L].repeat_interleave(2, 1), sin[:L].repeat_interleave(2, 1)
    rq = O.apply_rope(O.rms_norm(q, g, 1e-6), c2, s2).transpose(1, 2)
    dq, dk = (torch.zeros(B, H, S, 128, device=dev, dtype=torch.bfloat16) for _ in range(2))
    dv = torch.zeros(B, H, 128, pitch, device=dev, dtype=torch.bfloat16)
    rt.op_qkv_split(qkv.to(dev), L, H, [g.to(dev), g.to(dev), None], [pos.to(dev), pos.to(dev), None],
                    [dq, dk, dv], S, Lv, 1e-6, cos.to(dev), sin.to(dev), vt_pitch=pitch)
    assert rel_err(dq[:, :, Lv:].float(), rq) < 4e-3
    assert torch.equal(dv[..., Lv:Lv + L].cpu(), v.transpose(1, 2).transpose(2, 3).to(torch.bfloat16))
    assert float(dv[..., :Lv].abs().max()) == 0.0 and float(dv[..., S:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,L,H,Lv,tile", [(2, 11, 3, 4, 0), (2, 250, 2, 40, 0), (3, 70, 2, 3, 5), (2, 250, 1, 40, 9),
                                           (2, 250, 2, 40, 15), (3, 70, 1, 3, 19), (2, 250, 2, 40, 25), (3, 70, 1, 3, 29),
                                           (4, 37, 2, 8, 2), (2, 250, 2, 40, 27), (3, 70, 1, 3, 27), (2, 250, 2, 40, 26), (3, 70, 1, 3, 26), (2, 250, 2, 40, 28), (3, 70, 1, 3, 28),
                                           (2, 250, 2, 40, 32), (3, 70, 1, 3, 32), (5, 250, 3, 40, 32)])   # 32: the 256x256 BK = 32 tile, one head per 128-column epilogue pass (N = 384: a half-dead last tile)
def test_gemm_fused_head_split(dev, dtype, B, L, H, Lv, tile):
    """q/k/v projection with the head split fused into the GEMM epilogue (RMSNorm + RoPE into
    [B, H, S, 128]; bf16: V transposed [B, H, 128, pitch]) vs the unfused oracle math."""
    if tile in (9, 15, 19, 25, 26, 27, 28, 29, 32) and dtype == torch.float32:
        pytest.skip("bf16-only tile")
    K = 256
    S = L + Lv
    pitch = (S + 31) // 32 * 32
    x, w, b = _rand((B * L, K), 90), _rand((3 * H * 128, K), 91, 1 / math.sqrt(K)), _rand((3 * H * 128,), 92, 0.1)
    gq, gk = 1 + 0.1 * _rand((128,), 93), 1 + 0.1 * _rand((128,), 94)
    pos = (2 * torch.arange(L)).to(torch.int32)
    cos, sin = tables.rope_table(2 * L + 1)
    qkv = F.linear(_q(x, dtype), _q(w, dtype), b)
    q, k, v = qkv.view(B, L, 3, H, 128).unbind(2)
    c2, s2 = cos[pos.long()].repeat_interleave(2, 1), sin[pos.long()].repeat_interleave(2, 1)
    rq = O.apply_rope(O.rms_norm(q, gq, 1e-6), c2, s2).transpose(1, 2)
    rk = O.apply_rope(O.rms_norm(k, gk, 1e-6), c2, s2).transpose(1, 2)
    bf = dtype != torch.float32          # 16-bit operands: V leaves transposed
    dq, dk = (torch.zeros(B, H, S, 128, device=dev, dtype=dtype) for _ in range(2))
    dv = torch.zeros((B, H, 128, pitch) if bf else (B, H, S, 128), device=dev, dtype=dtype)
    desc = rt.qkv_split_desc(L, H, [gq.to(dev), gk.to(dev), None], [pos.to(dev), pos.to(dev), None], [dq, dk, dv], S, Lv,
                             1e-6, cos.to(dev), sin.to(dev), vt_pitch=pitch if bf else 0)
    rt.op_gemm(x.to(dev, dtype), w.to(dev, dtype), b.to(dev), epilogue=rt.EPI_QKV_SPLIT, qkv=desc, tile=tile)
    tol = {torch.float32: 1e-5, torch.bfloat16: 4e-3, torch.float16: 6e-4}[dtype]
    assert rel_err(dq[:, :, Lv:].float(), rq) < tol and rel_err(dk[:, :, Lv:].float(), rk) < tol
    assert float(dq[:, :, :Lv].float().abs().max()) == 0.0
    if bf:
        assert rel_err(dv[..., Lv:Lv + L].float(), v.transpose(1, 2).transpose(2, 3)) < tol
        assert float(dv[..., :Lv].float().abs().max()) == 0.0 and float(dv[..., S:].float().abs().max()) == 0.0
    else:
        assert rel_err(dv[:, :, Lv:], v.transpose(1, 2)) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("clips,L,H,Skv,tile", [(1, 250, 12, 77, 0), (1, 40, 12, 77, 0), (2, 50, 2, 77, 27), (3, 8, 2, 33, 27), (1, 250, 2, 96, 27),
                                                 (8, 250, 12, 77, 0)])
def test_gemm_cross_q_with_attention_epilogue(dev, dtype, clips, L, H, Skv, tile):
    """Cross-attention q projection with the attention against the cached text keys in its epilogue (hifi_foley.py:271-319;
    QkvSplitArgs::attn_*): rows [cfg = 2][clip][l], two text sets (one per CFG half) - tiles that straddle the halves run
    both.  Against the oracle math, and bit-identical to the unfused pair (head-split GEMM, then foley_op_attention) is
    not required (the unfused q is rounded once more on its way through memory - the same values here).  Large grids
    (last case) must report the plain head split and leave q in dst[0]."""
    K, D = 256, H * 128
    Bc, M = 2 * clips, 2 * clips * L
    pitch = 96
    x, w, b = _rand((M, K), 100), _rand((D, K), 101, 1 / math.sqrt(K)), _rand((D,), 102, 0.1)
    gq = 1 + 0.1 * _rand((128,), 103)
    pos = torch.arange(L).to(torch.int32)
    cos, sin = tables.rope_table(L + 1)
    kk, vv = _rand((2, H, Skv, 128), 104), _rand((2, H, Skv, 128), 105)
    q = F.linear(_q(x, dtype), _q(w, dtype), b).view(Bc, L, H, 128)
    c2, s2 = cos[pos.long()].repeat_interleave(2, 1), sin[pos.long()].repeat_interleave(2, 1)
    rq = _q(O.apply_rope(O.rms_norm(q, gq, 1e-6), c2, s2).transpose(1, 2), dtype)          # [Bc, H, L, 128], rounded like the kernel's
    sets = torch.arange(Bc) // clips
    ref = O.sdpa(rq, _q(kk, dtype)[sets], _q(vv, dtype)[sets]).transpose(1, 2).reshape(M, D)
    kd = kk.to(dev, dtype)
    vt = torch.zeros(2, H, 128, pitch, device=dev, dtype=dtype)
    vt[..., :Skv] = vv.transpose(2, 3).to(dev, dtype)
    dq = torch.zeros(Bc, H, L, 128, device=dev, dtype=dtype)
    out = torch.full((M, D), float("nan"), device=dev, dtype=dtype)
    desc = rt.qkv_split_desc(L, H, [gq.to(dev)], [pos.to(dev)], [dq], L, 0, 1e-6, cos.to(dev), sin.to(dev),
                             attn=(kd, vt, out, clips))
    rt.op_gemm(x.to(dev, dtype), w.to(dev, dtype), b.to(dev), epilogue=rt.EPI_QKV_SPLIT, qkv=desc, tile=tile)
    tol = {torch.bfloat16: 6e-3, torch.float16: 1e-3}[dtype]
    b128 = (M + 127) // 128 * H                 # the launcher's grid measure: the 64-row head-split tile serves 24 .. 100
    big = b128 > 100
    assert desc.fused() == (tile == 27 or 24 <= b128 <= 100)
    if desc.fused():
        assert float(dq.float().abs().max()) == 0.0                    # q never left the chip
        assert rel_err(out.float(), ref) < tol
    else:
        assert rel_err(dq.float(), rq) < tol and bool(torch.isnan(out.float()).all())
        rt.op_attention(dq, kd, vt, out, out, 0, kv_bdiv=clips)
        assert rel_err(out.float(), ref) < tol


def test_latent_rows(dev):
    x = _rand((3, 128, 50), 60)
    out = torch.empty(2 * 3 * 50, 128, device=dev, dtype=torch.bfloat16)
    rt.op_latent_rows(x.to(dev), 2, out)
    ref = x.transpose(1, 2).reshape(150, 128).to(torch.bfloat16)
    assert torch.equal(out.cpu()[:150], ref) and torch.equal(out.cpu()[150:], ref)


@pytest.mark.parametrize("solver", ["euler", "heun-2", "midpoint-2", "kutta-4"])
@pytest.mark.parametrize("ncfg", [1, 2])
def test_solver_step(dev, solver, ncfg):
    """Device solver state machine vs the oracle's restatement of FlowMatchDiscreteScheduler.step."""
    clips, C, L, n = 2, 128, 50, 8
    g = 4.5
    sig = O.flow_sigmas(n)
    st = O.SolverState(sig, solver)
    coef = tables.solver_table(sig, solver, n).to(dev)
    x0 = _rand((clips, C, L), 70)
    x = x0.to(dev).clone()
    xs, da = torch.zeros_like(x), torch.zeros_like(x)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    rows = torch.empty(ncfg * clips * L, C, device=dev)
    xr = x0.clone()
    for i in range(n):
        pred = _rand((ncfg * clips * L, C), 71 + i)
        rt.op_solver_step(pred.to(dev), x, xs, da, ncfg, g, coef, ctr, rows)
        p = pred.view(ncfg, clips, L, C).transpose(2, 3)
        v = p[0] + g * (p[1] - p[0]) if ncfg == 2 else p[0]
        xr = st.step(v, xr)
        assert rel_err(x, xr) < 2e-6, (solver, i)
        assert rel_err(rows.view(ncfg, clips, L, C)[-1], xr.transpose(1, 2)) < 2e-6
    assert int(ctr.item()) == n


def test_dac_out(dev):
    B, T, C = 2, 300, 64
    s, w, b = _rand((B, T, C), 80), _rand((1, C, 7), 81, 0.1), _rand((1,), 82, 0.1)
    ref = torch.tanh(F.conv1d(s.transpose(1, 2), w, b, padding=3))
    out = torch.empty(B, 1, T, device=dev)
    rt.op_dac_out(s.to(dev), w[0].permute(1, 0).reshape(-1).contiguous().to(dev), b.to(dev), out)
    assert rel_err(out, ref) < 2e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,H,Sq,Skv,rows", [(3, 12, 8, 9, 40), (2, 12, 1, 197, 400), (5, 4, 196, 197, 1000), (1, 12, 130, 130, 130)])
def test_qkv_regroup(dev, dtype, G, H, Sq, Skv, rows):
    """foley_op_qkv_regroup (the conditioning encoders' token regrouping: vit_helper.py:37-105 DividedAttention's rearranges + CLS
    prepend, the HF encoders' head split) against the torch expression it replaces: exact copies; 16-bit operands get V transposed
    with a zero pad up to ceil32(Skv).  Lengths off the 64-token tile, Sq != Skv, a single query, repeated source rows."""
    from foley_amd.host import runtime as rt
    g = torch.Generator().manual_seed(G * 1000 + Sq)
    qkv = torch.randn(rows, 3 * H * 64, generator=g).to(dtype).to(dev)
    iq = torch.randint(0, rows, (G, Sq), generator=g, dtype=torch.int32).to(dev)
    ikv = torch.randint(0, rows, (G, Skv), generator=g, dtype=torch.int32).to(dev)
    q, k, v = rt.op_qkv_regroup(qkv, H, iq, ikv)
    t = qkv.view(rows, 3, H, 64)
    want_q = t[iq.long(), 0].permute(0, 2, 1, 3)          # [G, H, Sq, 64]
    want_k = t[ikv.long(), 1].permute(0, 2, 1, 3)
    want_v = t[ikv.long(), 2].permute(0, 2, 1, 3)
    assert torch.equal(q, want_q) and torch.equal(k, want_k)
    if dtype == torch.float32:
        assert torch.equal(v, want_v)
    else:
        pitch = (Skv + 31) // 32 * 32
        assert v.shape == (G, H, 64, pitch)
        assert torch.equal(v[..., :Skv], want_v.transpose(2, 3)) and not bool(v[..., Skv:].any())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,H,Sq,Skv", [(3, 12, 8, 9), (2, 12, 1, 197), (5, 4, 196, 197), (1, 12, 130, 77)])
def test_attention_scatter(dev, dtype, G, H, Sq, Skv):
    """foley_op_attention_scatter: the head_dim-64 attention whose query (g, t) is written to row out_rows[g, t] of a token-major
    buffer (the conditioning encoders: the inverse rearrange + torch.cat of vit_helper.py:95-105) - EQUAL to the plain
    foley_op_attention_hd output placed by the same permutation; rows no query maps to stay untouched."""
    from foley_amd.host import runtime as rt
    g = torch.Generator().manual_seed(G * 100 + Sq)
    rows = G * Sq + 7
    qkv = torch.randn(rows, 3 * H * 64, generator=g).to(dtype).to(dev)
    iq = torch.randperm(rows, generator=g)[: G * Sq].view(G, Sq).to(torch.int32).to(dev)
    ikv = torch.randint(0, rows, (G, Skv), generator=g, dtype=torch.int32).to(dev)
    q, k, v = rt.op_qkv_regroup(qkv, H, iq, ikv)
    plain = torch.empty(G, Sq, H * 64, device=dev, dtype=dtype)
    rt.op_attention(q, k, v, plain, plain, 0)
    out = torch.full((rows, H * 64), 7.0, device=dev, dtype=dtype)
    rt.op_attention_scatter(q, k, v, iq, out)
    want = torch.full((rows, H * 64), 7.0, device=dev, dtype=dtype)
    want[iq.view(-1).long()] = plain.view(G * Sq, H * 64)
    assert torch.equal(out, want)
    with pytest.raises(rt.FoleyRuntimeError):
        rt.op_attention_scatter(q, k, v, iq.long(), out)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,H,p,gq,gkv", [(6, 12, 14, 8, 9), (3, 4, 16, 8, 9), (2, 12, 5, 8, 9), (4, 2, 3, 33, 40), (1, 12, 1, 8, 9)])
def test_attention_scatter_grouped(dev, dtype, G, H, p, gq, gkv):
    """The block-diagonal form of foley_op_attention_scatter (grp_q / grp_kv): p small groups packed into one sequence must give
    what the same groups give one by one (the plain kernel, one group per sequence) - the Synchformer's time attention (8 frame
    queries x (CLS + 8) keys, 14 locations per workgroup), packs that end inside a 32-key tile, groups that straddle tiles, a wave
    whose queries see no key of a tile, idle waves, p = 1."""
    from foley_amd.host import runtime as rt
    g = torch.Generator().manual_seed(G * 100 + p)
    n = G * p                      # small groups
    rows = n * gkv + 5
    qkv = torch.randn(rows, 3 * H * 64, generator=g).to(dtype).to(dev)
    iq = torch.randperm(rows, generator=g)[: n * gq].view(n, gq).to(torch.int32).to(dev)
    ikv = torch.randint(0, rows, (n, gkv), generator=g, dtype=torch.int32).to(dev)
    q, k, v = rt.op_qkv_regroup(qkv, H, iq, ikv)                    # one group per sequence
    want = torch.zeros(rows, H * 64, device=dev, dtype=dtype)
    rt.op_attention_scatter(q, k, v, iq, want)
    qp, kp, vp = rt.op_qkv_regroup(qkv, H, iq.view(G, p * gq), ikv.view(G, p * gkv))   # p groups per sequence
    out = torch.zeros(rows, H * 64, device=dev, dtype=dtype)
    rt.op_attention_scatter(qp, kp, vp, iq.view(G, p * gq), out, gq, gkv)
    err = float((out.float() - want.float()).abs().max())
    assert bool(torch.isfinite(out.float()).all()) and err <= (2e-2 if dtype == torch.bfloat16 else 3e-3), err
    with pytest.raises(rt.FoleyRuntimeError):       # the last group's keys must lie inside Skv
        rt.op_attention_scatter(qp, kp, vp, iq.view(G, p * gq), out, gq, gkv + 1)
