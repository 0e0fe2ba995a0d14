"""GPU parity, operator level: every C-ABI entry point against a plain PyTorch fp32 (CPU)
restatement of the same reference op, on the same seeded inputs.

Inputs / weights are first rounded to the engine's 16-bit storage type so the comparison
isolates the kernel (accumulation order, fp32 statistics, output rounding).  Tolerances are
relative L2 and are stated per dtype:  fp16 2e-3, bf16 1.5e-2 for 16-bit outputs
(unit round-off 4.9e-4 / 3.9e-3), 2e-5 for fp32 outputs of fp32 data paths, bit-exact for
index-only and un-fused fp32 sampler arithmetic.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import weights as W

pytestmark = pytest.mark.gpu

DT = [torch.float16, torch.bfloat16]
# rel-L2 of one kernel against fp32 torch on inputs pre-rounded to the storage type: 2x the largest value measured on the
# MI355X over every case of this file (fp16 2.5e-4, bf16 2.0e-3: profiles/r05_error_table.txt)
TOL = {torch.float16: 5e-4, torch.bfloat16: 4e-3}


def rel(a, b):
    from tests.golden_cases import record
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = float((a - b).norm() / b.norm().clamp_min(1e-30))
    record("rel", err)
    return err


def _k_orders():
    """The chunk-major k order of the packed weights (measured 10-15 % slower, kept for study) exists in the development
    build only (-DMOBI_DEV); the shipped library answers MOBI_ERR_UNSUPPORTED."""
    from mobi_amd import _lib
    return (False, True) if _lib.load().mobi_build_info() & 1 else (False,)


def rnd(name, shape, dtype, scale=1.0):
    """deterministic input, rounded to the storage type; returns (fp32 cpu copy, device tensor)."""
    x = (W.synth_input(name, shape) * scale).to(dtype)
    return x.float(), x.cuda()


@pytest.fixture(scope="module")
def ops():
    from mobi_amd import ops as o
    return o


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c0,c1,hw,silu", [(32, 0, 64, True), (320, 0, 4096, True), (320, 0, 4096, False),
                                           (1280, 1280, 64, True), (1280, 640, 256, True), (640, 320, 1024, True),
                                           (64, 0, 17, True), (128, 0, 4096, True), (1280, 0, 256, True),
                                           (1920, 640, 64, False), (960, 960, 16, True), (640, 0, 1024, True),
                                           (320, 0, 1024, True), (640, 0, 100, True), (1280, 0, 77, False),
                                           (1280, 640, 1024, True), (320, 320, 4096, True)])
def test_groupnorm(ops, dtype, c0, c1, hw, silu, tune):
    """The three forms of GroupNorm against fp32 torch: one launch with the slab in registers (gn_regs_kernel: 16- and
    8-byte pieces, segments that straddle the two sources, ragged pixel counts), one launch with the slab in LDS
    (gn_fused_kernel, MOBI_GN_FUSED=1) and the two-launch form (statistics, apply; MOBI_GN_FUSED=0 and every shape the
    one-launch forms do not take)."""
    h = int(math.isqrt(hw)) if int(math.isqrt(hw)) ** 2 == hw else 1
    w = hw // h
    xf, xd = rnd(f"gn{c0}.{c1}.{hw}", (2, h, w, c0), dtype, 2.0)
    xf = xf + 0.7                                            # non-zero mean: exercises the variance formula
    xd = xf.to(dtype).cuda()
    xf = xd.float().cpu()
    x2f = x2d = None
    if c1:
        x2f, x2d = rnd(f"gn2{c0}.{c1}.{hw}", (2, h, w, c1), dtype)
    C = c0 + c1
    g = torch.from_numpy(W.synth_param("g.weight", (C,)))
    b = torch.from_numpy(W.synth_param("g.bias", (C,)))
    y = ops.groupnorm(xd, g.cuda(), b.cuda(), 1e-5, silu, x2=x2d)
    ref_in = xf if x2f is None else torch.cat([xf, x2f], dim=3)
    ref = F.group_norm(ref_in.permute(0, 3, 1, 2), 32, g, b, 1e-5)
    ref = F.silu(ref) if silu else ref
    assert y.shape == (2, h, w, C)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[dtype]
    tune.setenv("MOBI_GN_FUSED", "0")                        # the two-launch form on the shapes the fused kernel takes
    y2 = ops.groupnorm(xd, g.cuda(), b.cuda(), 1e-5, silu, x2=x2d)
    assert rel(y2.float().permute(0, 3, 1, 2), ref) < TOL[dtype]
    tune.setenv("MOBI_GN_FUSED", "1")
    y3 = ops.groupnorm(xd, g.cuda(), b.cuda(), 1e-5, silu, x2=x2d)
    assert rel(y3.float().permute(0, 3, 1, 2), ref) < TOL[dtype]
    tune.delenv("MOBI_GN_FUSED")
    tune.setenv("MOBI_GN_COOP", "1")                         # pixel chunks meeting through memory, wherever the geometry fits
    for _ in range(3):                                       # (the arrival counters return to zero: the same buffer every time)
        y4 = ops.groupnorm(xd, g.cuda(), b.cuda(), 1e-5, silu, x2=x2d)
        assert rel(y4.float().permute(0, 3, 1, 2), ref) < TOL[dtype]
    torch.cuda.synchronize()
    for buf in ops._SYNC.values():
        assert int(buf.abs().sum()) == 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,c,hw", [(16, 320, 4096), (16, 640, 1024), (16, 1280, 256), (8, 320, 1024), (3, 640, 200)])
def test_groupnorm_chunks_meet_through_memory(ops, dtype, n, c, hw, tune):
    """gn_coop_kernel at the step's batch sizes (a grid of up to one block per CU, 16 / 32 chunks per image): a block owns a
    chunk of pixels x all channels, the chunks of an image exchange their partial sums through memory (device-coherent stores /
    loads, an arrival counter, a bounded spin).  Against fp32 torch and, bit for bit, against itself on repetition."""
    xf, xd = rnd(f"gnc{c}.{hw}", (n, 1, hw, c), dtype, 2.0)
    xd = (xf + 0.4).to(dtype).cuda()
    xf = xd.float().cpu()
    g = torch.from_numpy(W.synth_param("g.weight", (c,)))
    b = torch.from_numpy(W.synth_param("g.bias", (c,)))
    ref = F.silu(F.group_norm(xf.permute(0, 3, 1, 2), 32, g, b, 1e-5))
    tune.setenv("MOBI_GN_COOP", "1")
    ys = [ops.groupnorm(xd, g.cuda(), b.cuda(), 1e-5, True) for _ in range(4)]
    assert rel(ys[0].float().permute(0, 3, 1, 2), ref) < TOL[dtype]
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    torch.cuda.synchronize()
    for buf in ops._SYNC.values():
        assert int(buf.abs().sum()) == 0


@pytest.mark.parametrize("dtype", DT)
def test_groupnorm_fp32_source_and_precise_outputs(ops, dtype):
    """mobi_groupnorm_params.src_f32 / out_mode (the VAE decoder's fp32 streams and the lidar tail's hi | lo operands,
    ldm/modules/diffusionmodules/model.py): fp32 in, storage-type out; hi | lo pair (hi + lo = the fp32 result to ~2^-22); fp32 out."""
    n, hw, c = 2, 1024, 128
    x = (W.synth_input("gn32.x", (n, 32, 32, c)) * 1.7 + 0.3).cuda()
    g = torch.from_numpy(W.synth_param("g.weight", (c,)))
    b = torch.from_numpy(W.synth_param("g.bias", (c,)))
    ref = F.silu(F.group_norm(x.cpu().double().permute(0, 3, 1, 2), 32, g.double(), b.double(), 1e-6)).permute(0, 2, 3, 1)
    y = ops.groupnorm(x, g.cuda(), b.cuda(), 1e-6, True, dtype=dtype)
    assert y.dtype == dtype and rel(y.float(), ref.float()) < TOL[dtype]
    y32 = ops.groupnorm(x, g.cuda(), b.cuda(), 1e-6, True, out_mode=ops.GN_OUT_F32, dtype=dtype)
    assert y32.dtype == torch.float32 and rel(y32, ref.float()) < 2e-6
    pair = ops.groupnorm(x, g.cuda(), b.cuda(), 1e-6, True, out_mode=ops.GN_OUT_SPLIT, dtype=dtype)
    assert pair.shape == (n, 32, 32, 2 * c) and pair.dtype == dtype
    assert torch.equal(pair[..., :c], y32.to(dtype))                          # hi = T(y)
    both = pair[..., :c].double() + pair[..., c:].double()
    assert rel(both.float(), y32) < (2e-6 if dtype == torch.float16 else 2e-5)  # hi + lo: 22 (fp16) / 16 (bf16) bits of y
    # hi | lo | hi (for weights [W ; W ; W - T(W)]) and the same operand forms of a tensor no GroupNorm stands in front of
    tri = ops.groupnorm(x, g.cuda(), b.cuda(), 1e-6, True, out_mode=ops.GN_OUT_SPLIT3, dtype=dtype)
    assert tri.shape == (n, 32, 32, 3 * c) and torch.equal(tri[..., :2 * c], pair) and torch.equal(tri[..., 2 * c:], pair[..., :c])
    for parts in (2, 3):
        sp = ops.split_f32(y32, dtype, parts)
        assert sp.shape == (n, 32, 32, parts * c) and torch.equal(sp[..., :c], y32.to(dtype))
        assert torch.equal(sp[..., c:2 * c], (y32 - y32.to(dtype).float()).to(dtype))
        assert parts == 2 or torch.equal(sp[..., 2 * c:], sp[..., :c])
    # a convolution on the split operands against fp64: both operands to ~22 bits (fp16) where the plain launch has 11
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(c, 64, 3, padding=1)
    from mobi_amd.ldm.modules.diffusionmodules.util import Conv2d
    mine = Conv2d(c, 64, 3, padding=1)
    mine.load_state_dict(conv.state_dict())
    mine = mine.cuda()
    import mobi_amd
    was = mobi_amd.engine_dtype()
    mobi_amd.set_engine_dtype(dtype)
    try:
        want = F.conv2d(y32.cpu().double().permute(0, 3, 1, 2), conv.weight.double(), conv.bias.double(), padding=1).permute(0, 2, 3, 1)
        e_plain = rel(ops.igemm(y32.to(dtype), mine.packed(), out_mode=ops.OUT_ROWS_F32), want.float())
        e_pair = rel(ops.igemm(pair, mine.packed_dup(), out_mode=ops.OUT_ROWS_F32), want.float())
        e_tri = rel(ops.igemm(tri, mine.packed_dup3(), out_mode=ops.OUT_ROWS_F32), want.float())
    finally:
        mobi_amd.set_engine_dtype(was)
    assert e_tri < 0.2 * e_pair < 0.2 * e_plain, (e_plain, e_pair, e_tri)
    assert e_tri < (3e-6 if dtype == torch.float16 else 3e-4), e_tri
    # the storage-type source through the same kernels
    xs = x.to(dtype)
    p2 = ops.groupnorm(xs, g.cuda(), b.cuda(), 1e-6, True, out_mode=ops.GN_OUT_F32)
    ref2 = F.silu(F.group_norm(xs.cpu().double().permute(0, 3, 1, 2), 32, g.double(), b.double(), 1e-6)).permute(0, 2, 3, 1)
    assert rel(p2, ref2.float()) < 2e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,t,strided", [(64, 16, False), (320, 4096, False), (640, 100, True), (1280, 64, True)])
def test_layernorm(ops, dtype, c, t, strided):
    xf, xd = rnd(f"ln{c}", (4, t, c), dtype, 1.5)
    g = torch.from_numpy(W.synth_param("ln.weight", (c,)))
    b = torch.from_numpy(W.synth_param("ln.bias", (c,)))
    if strided:
        y = ops.layernorm(xd[1::2], g.cuda(), b.cuda())
        ref = F.layer_norm(xf[1::2], (c,), g, b, 1e-5)
    else:
        y = ops.layernorm(xd, g.cuda(), b.cuda())
        ref = F.layer_norm(xf, (c,), g, b, 1e-5)
    assert rel(y.float(), ref) < TOL[dtype]


# ---------------------------------------------------------------------------------------------
def _conv_ref(xf, wf, bias, stride=1, pad=(1, 1), upsample=False, asym=False):
    x = xf.permute(0, 3, 1, 2)
    if upsample:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if asym:
        x = F.pad(x, (0, 1, 0, 1))
        pad = (0, 0)
    return F.conv2d(x, wf, bias, stride=stride, padding=pad).permute(0, 2, 3, 1)


IGEMM_CASES = [
    # name, cin, cout, kh, kw, h, w, stride, upsample, asym
    ("lin320", 320, 960, 1, 1, 64, 1, 1, False, False),
    ("c3_64_160", 64, 160, 3, 3, 16, 16, 1, False, False),
    ("c3_320_320", 320, 320, 3, 3, 32, 32, 1, False, False),
    ("c3_s2", 64, 64, 3, 3, 16, 16, 2, False, False),
    ("c3_up", 64, 128, 3, 3, 8, 8, 1, True, False),
    ("c3_asym", 32, 32, 3, 3, 16, 16, 2, False, True),
    ("c15", 32, 64, 1, 5, 8, 24, 1, False, False),
    ("c3_odd", 96, 40, 3, 3, 7, 9, 1, False, False),
    ("lin_k32", 32, 24, 1, 1, 5, 1, 1, False, False),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", IGEMM_CASES, ids=[c[0] for c in IGEMM_CASES])
def test_igemm_conv(ops, dtype, case):
    name, cin, cout, kh, kw, h, w, stride, up, asym = case
    xf, xd = rnd("x." + name, (3, h, w, cin), dtype)
    wf = torch.from_numpy(W.synth_param(name + ".weight", (cout, cin, kh, kw))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param(name + ".bias", (cout,)))
    pw = ops.pack_conv(wf, bias, dtype, "cuda")
    pad = (kh // 2, kw // 2)
    if asym:
        y = ops.igemm(xd, pw, stride=2, pad=(0, 0), hout=(h + 1 - 3) // 2 + 1, wout=(w + 1 - 3) // 2 + 1)
    else:
        y = ops.igemm(xd, pw, stride=stride, pad=pad, upsample=up)
    ref = _conv_ref(xf, wf, bias, stride, pad, up, asym)
    assert y.shape == ref.shape
    assert rel(y.float(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_igemm_epilogues(ops, dtype):
    """concat of two sources + per-image vector + residual; fp32 output; transposed output."""
    x0f, x0d = rnd("e.x0", (4, 8, 8, 64), dtype)
    x1f, x1d = rnd("e.x1", (4, 8, 8, 32), dtype)
    rf, rd = rnd("e.res", (4, 8, 8, 160), dtype)
    wf = torch.from_numpy(W.synth_param("e.weight", (160, 96, 3, 3))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param("e.bias", (160,)))
    rv = W.synth_input("e.rowvec", (4, 200))[:, 20:180].contiguous()
    rvd = W.synth_input("e.rowvec", (4, 200)).cuda()[:, 20:180]               # strided rows
    pw = ops.pack_conv(wf, bias, dtype, "cuda")
    ref = _conv_ref(torch.cat([x0f, x1f], 3), wf, bias) + rv[:, None, None, :] + rf
    y = ops.igemm(x0d, pw, x2=x1d, rowvec=rvd, residual=rd)
    assert rel(y.float(), ref) < TOL[dtype]
    from mobi_amd._lib import OUT_ROWS_F32, OUT_TRANSPOSED
    ref2 = _conv_ref(torch.cat([x0f, x1f], 3), wf, bias)
    y32 = ops.igemm(x0d, pw, x2=x1d, out_mode=OUT_ROWS_F32, scale=0.5)
    ref_s = _conv_ref(torch.cat([x0f, x1f], 3), wf, None) * 0.5 + bias
    assert y32.dtype == torch.float32 and rel(y32, ref_s) < 2e-5 * (100 if dtype == torch.bfloat16 else 1) + 1e-6
    yt = ops.igemm(x0d, pw, x2=x1d, out_mode=OUT_TRANSPOSED)
    assert yt.shape == (4, 160, 64)
    assert rel(yt.float(), ref2.reshape(4, 64, 160).permute(0, 2, 1)) < TOL[dtype]
    # transposed with a spatial size that is not a multiple of 8 (scalar store path)
    xs_f, xs_d = rnd("e.xs", (3, 1, 3, 32), dtype)
    ws = torch.from_numpy(W.synth_param("e.ws", (40, 32, 1, 1))).to(dtype).float()
    pws = ops.pack_conv(ws, None, dtype, "cuda")
    yt = ops.igemm(xs_d, pws, out_mode=OUT_TRANSPOSED)
    assert rel(yt.float(), _conv_ref(xs_f, ws, None, pad=(0, 0)).reshape(3, 3, 40).permute(0, 2, 1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c", [320, 64, 96])
def test_igemm_geglu_and_inplace_strided(ops, dtype, c):
    xf, xd = rnd(f"g.x{c}", (4, 50, c), dtype)
    wf = torch.from_numpy(W.synth_param(f"g{c}.weight", (8 * c, c))).to(dtype).float()
    bf = torch.from_numpy(W.synth_param(f"g{c}.bias", (8 * c,)))
    y = ops.linear(xd, ops.pack_geglu(wf, bf, dtype, "cuda"))
    a, gate = F.linear(xf, wf, bf).chunk(2, dim=-1)
    assert y.shape == (4, 50, 4 * c)
    assert rel(y.float(), a * F.gelu(gate)) < TOL[dtype]
    # camera/lidar halves updated in place through batch-strided views (attention.py:245-263)
    w2 = torch.from_numpy(W.synth_param(f"g2{c}.weight", (c, c))).to(dtype).float()
    b2 = torch.from_numpy(W.synth_param(f"g2{c}.bias", (c,)))
    pw2 = ops.pack_linear(w2, b2, dtype, "cuda")
    af, ad = rnd(f"g.a{c}", (2, 50, c), dtype)
    xd2 = xd.clone()
    ops.linear(ad, pw2, residual=xd2[::2], out=xd2[::2])
    ref = xf.clone()
    ref[::2] = F.linear(af, w2, b2) + xf[::2]
    assert rel(xd2.float(), ref) < TOL[dtype]
    assert torch.equal(xd2[1::2], xd[1::2])                       # the other half is untouched


@pytest.mark.parametrize("dtype", DT)
def test_igemm_per_image_weights(ops, dtype):
    """S = q k^T / sqrt(c) and O = P v with one weight matrix per image (VAE AttnBlock)."""
    from mobi_amd._lib import OUT_ROWS_F32
    n, t, c = 2, 64, 64
    qf, qd = rnd("pi.q", (n, t, 1, c), dtype)
    kf, kd = rnd("pi.k", (n, t, c), dtype)
    s = ops.igemm(qd, ops.Packed(kd, None, 1, 1, c, t, t), weight_per_image=True, w_group_stride=t * c,
                  out_mode=OUT_ROWS_F32, scale=c ** -0.5)
    ref = torch.einsum("ntc,nsc->nts", qf[:, :, 0], kf) * c ** -0.5
    assert rel(s.view(n, t, t), ref) < 1e-3
    p = ops.softmax_rows(s.view(n * t, t), dtype)
    assert rel(p.float().view(n, t, t), torch.softmax(s.view(n, t, t).cpu(), dim=-1)) < TOL[dtype]


# ---------------------------------------------------------------------------------------------
ATTN_CASES = [(8, 8, 64, 64), (8, 16, 100, 100), (8, 40, 256, 256), (8, 80, 64, 64), (8, 160, 64, 64),
              (4, 32, 1, 1), (8, 8, 4, 4), (8, 40, 4096, 4096), (2, 64, 70, 130), (8, 24, 16, 16)]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("v_rows", [True, False], ids=["v_rows", "v_transposed"])
@pytest.mark.parametrize("heads,dh,tq,tk", ATTN_CASES)
def test_attention(ops, dtype, heads, dh, tq, tk, v_rows, tune):
    """both V layouts of the C ABI: row-major V (transposing LDS reads) and pre-transposed V^T; the row-major path also
    with 8-wave blocks (256 queries per staged K / V tile), which the library otherwise picks for big launches only."""
    n, c = 2, heads * dh
    qf, qd = rnd(f"a.q{dh}.{tq}", (n, tq, c), dtype, 1.5)
    kf, kd = rnd(f"a.k{dh}.{tk}", (n, tk, c), dtype, 1.5)
    vf, vd = rnd(f"a.v{dh}.{tk}", (n, tk, c), dtype)
    vt = vd if v_rows else vd.permute(0, 2, 1).contiguous()
    y = ops.attention(qd, kd, vt, heads, dh ** -0.5, v_rows=v_rows)
    sp = lambda t: t.reshape(n, -1, heads, dh).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", sp(qf), sp(kf)) * dh ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), sp(vf)).permute(0, 2, 1, 3).reshape(n, tq, c)
    assert rel(y.float(), ref) < TOL[dtype]
    # q already carrying scale * log2(e) (what the transformer blocks do: folded into the packed to_q weights)
    log2e = 1.4426950408889634
    qsf = (qf * (dh ** -0.5 * log2e)).to(dtype)
    sim2 = torch.einsum("bhid,bhjd->bhij", sp(qsf.float()), sp(kf)) / log2e
    ref2 = torch.einsum("bhij,bhjd->bhid", sim2.softmax(-1), sp(vf)).permute(0, 2, 1, 3).reshape(n, tq, c)
    y2 = ops.attention(qsf.cuda(), kd, vt, heads, 123.0, v_rows=v_rows, q_log2_scaled=True)
    assert rel(y2.float(), ref2) < TOL[dtype]
    if v_rows and dh <= 80:
        if dh == 40:
            # the A/B partner that runs the last (3/4 padded) channel block's P.V on MFMA 16x16x32 (MOBI_ATTN_H16=1), both block sizes
            tune.setenv("MOBI_ATTN_H16", "1")
            for nw in ("4", "8"):
                tune.setenv("MOBI_ATTN_NW", nw)
                yh = ops.attention(qd, kd, vt, heads, dh ** -0.5, v_rows=v_rows)
                assert rel(yh.float(), ref) < TOL[dtype], nw
            tune.delenv("MOBI_ATTN_H16")
            tune.delenv("MOBI_ATTN_NW")
        tune.setenv("MOBI_ATTN_NW", "8")
        y8 = ops.attention(qd, kd, vt, heads, dh ** -0.5, v_rows=v_rows)
        assert rel(y8.float(), ref) < TOL[dtype]
        # the kernel these launches ran on before attention_rows_kernel (still the one for V^T inputs and dh > 80)
        tune.setenv("MOBI_ATTN_V3", "0")
        for nw in ("4", "8"):
            tune.setenv("MOBI_ATTN_NW", nw)
            y0 = ops.attention(qd, kd, vt, heads, dh ** -0.5, v_rows=v_rows)
            assert rel(y0.float(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nw", ["4", "8"])
@pytest.mark.parametrize("dh,tq,tk,kind", [(40, 300, 1000, "ramp"), (40, 256, 4096, "ramp"), (80, 130, 700, "ramp"),
                                           (64, 70, 390, "ramp"), (48, 64, 512, "ramp"), (40, 96, 640, "huge"),
                                           (80, 64, 320, "huge"), (32, 64, 256, "huge"), (40, 64, 200, "negative"),
                                           (8, 40, 130, "ramp"), (24, 33, 257, "spike"), (40, 128, 4096, "peaky")])
def test_attention_shift_path(ops, dtype, nw, dh, tq, tk, kind, tune):
    """attention_rows_kernel keeps a shift below the running maximum and raises it only when a probability reaches 2.0
    (the OR test on the packed words); these inputs make that exact path run often or at the extremes: scores that keep
    rising along the keys ('ramp'), scores of magnitude ~1e3 ('huge': the speculative exp2 overflows), all scores far
    below zero ('negative': the first tile must set the shift), one late dominant key ('spike')."""
    n, heads = 2, 4
    c = heads * dh
    qf, _ = rnd(f"as.q{dh}.{tq}", (n, tq, c), dtype, 1.5)
    kf, _ = rnd(f"as.k{dh}.{tk}", (n, tk, c), dtype, 1.5)
    vf, vd = rnd(f"as.v{dh}.{tk}", (n, tk, c), dtype)
    if kind == "ramp":
        kf = kf * torch.linspace(0.2, 6.0, tk).view(1, tk, 1)
    elif kind == "huge":
        kf = kf * 60.0
        qf = qf * 8.0
    elif kind == "negative":
        qf = qf.abs() + 1.0
        kf = -(kf.abs() + 1.0) * 4.0
    elif kind == "spike":
        kf[:, tk - 3] = qf[:, 0:1].mean(1) * 9.0
    elif kind == "peaky":
        # a peaky distribution at the production key count (T = 4,096, dh = 40): every query has a few keys 15 .. 25 nats above
        # the rest, some of them late in the row (beyond the 32 keys whose maximum fixes the shift on padded head dims; fp16
        # keeps 13.9 nats of headroom above it: these rows take the block's second, tested pass)
        for j, pos in enumerate((5, 700, 2049, 4090)):
            kf[:, pos] = qf[:, j::4].mean(1) * (3.0 + j) + kf[:, pos] * 0.2
    qd, kd = qf.to(dtype).cuda(), kf.to(dtype).cuda()
    qf, kf = qd.float().cpu(), kd.float().cpu()
    sp = lambda t: t.reshape(n, -1, heads, dh).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", sp(qf).double(), sp(kf).double()) * dh ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), sp(vf).double()).permute(0, 2, 1, 3).reshape(n, tq, c)
    tune.setenv("MOBI_ATTN_NW", nw)
    y = ops.attention(qd, kd, vd, heads, dh ** -0.5, v_rows=True)
    assert torch.isfinite(y.float()).all()
    # the kernel rounds Q' = Q * scale * log2(e) to the storage type once more (callers that fold the factor into to_q do
    # not pay this: q_log2_scaled): 1.5 x the one-kernel bound; scores of ~1e3 ('huge') move by several units with that
    # rounding -- measured on the MI355X 7.4e-3 (fp16) / 5.9e-2 (bf16), asserted at 2x
    tol = {"huge": {torch.float16: 1.5e-2, torch.bfloat16: 1.2e-1}[dtype]}.get(
        kind, TOL[dtype] * (2.5 if kind in ("ramp", "spike", "peaky") else 1.5))
    assert rel(y.float(), ref) < tol
    if dh == 40:
        # the same inputs through the 16x16x32 form of the last channel block (MOBI_ATTN_H16=1: ragged last tile, raised shifts,
        # the block's second pass)
        tune.setenv("MOBI_ATTN_H16", "1")
        yh = ops.attention(qd, kd, vd, heads, dh ** -0.5, v_rows=True)
        tune.delenv("MOBI_ATTN_H16")
        assert torch.isfinite(yh.float()).all() and rel(yh.float(), ref) < tol
    if kind in ("huge", "peaky"):
        # the form the transformer blocks use -- Q' handed over already scaled (q_log2_scaled), so the kernel adds no
        # rounding of its own -- against a reference built from the SAME rounded Q': the one-kernel bound holds on the very
        # inputs that overflow the speculative exponentials
        log2e = 1.4426950408889634
        qs = (qf * (dh ** -0.5 * log2e)).to(dtype)
        sim2 = torch.einsum("bhid,bhjd->bhij", sp(qs.float()).double(), sp(kf).double()) / log2e
        ref2 = torch.einsum("bhij,bhjd->bhid", sim2.softmax(-1), sp(vf).double()).permute(0, 2, 1, 3).reshape(n, tq, c)
        y2 = ops.attention(qs.cuda(), kd, vd, heads, 1.0, v_rows=True, q_log2_scaled=True)
        assert torch.isfinite(y2.float()).all()
        assert rel(y2.float(), ref2) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("dh,tq,tk", [(40, 300, 40), (40, 1, 64), (48, 257, 100), (40, 256, 128), (40, 70, 190),
                                      (40, 512, 300), (40, 33, 1)])
def test_attention_software_pipelined(ops, dtype, dh, tq, tk, tune):
    from mobi_amd import _lib
    if not _lib.load().mobi_build_info() & 1:
        pytest.skip("attention_sp_kernel is an A/B kernel of the development build (-DMOBI_DEV), not in the shipped library")
    """the software-pipelined 8-wave kernel (head dims 33..48) on 1..5 key tiles, ragged last tiles and ragged query blocks;
    it is an A/B alternative (MOBI_ATTN_SP=1, slower than the default kernel) and must agree with the default kernel."""
    n, heads = 2, 4
    c = heads * dh
    qf, qd = rnd(f"ap.q{dh}.{tq}", (n, tq, c), dtype, 1.5)
    kf, kd = rnd(f"ap.k{dh}.{tk}", (n, tk, c), dtype, 1.5)
    vf, vd = rnd(f"ap.v{dh}.{tk}", (n, tk, c), dtype)
    sp = lambda t: t.reshape(n, -1, heads, dh).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", sp(qf), sp(kf)) * dh ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), sp(vf)).permute(0, 2, 1, 3).reshape(n, tq, c)
    tune.setenv("MOBI_ATTN_NW", "8")
    tune.setenv("MOBI_ATTN_SP", "1")
    y = ops.attention(qd, kd, vd, heads, dh ** -0.5, v_rows=True)
    assert torch.isfinite(y.float()).all()
    assert rel(y.float(), ref) < TOL[dtype]
    tune.setenv("MOBI_ATTN_SP", "0")
    y0 = ops.attention(qd, kd, vd, heads, dh ** -0.5, v_rows=True)
    assert rel(y.float(), y0.float()) < 1e-6 + (0 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", DT)
def test_attention_strided_partner_and_spike(ops, dtype, tune):
    """q from the camera half, k/v from the lidar half of an interleaved batch (image strides), with
    q/k packed in one [.., 2C] tensor (row stride > C) and one huge score (online-softmax rescale)."""
    n, t, heads, dh = 4, 130, 8, 40
    c = heads * dh
    xf, xd = rnd("as.qk", (n, t, 2 * c), dtype)
    xf[0, 5, :dh] *= 6.0
    xf[1, 77, c:c + dh] = xf[0, 5, :dh]                     # key 77 of the partner matches query 5
    xd = xf.to(dtype).cuda()
    xf = xd.float().cpu()
    vf, vd = rnd("as.v", (n, t, c), dtype)
    vt = vd.permute(0, 2, 1).contiguous()
    y = ops.attention(xd[::2, :, :c], xd[1::2, :, c:], vt[1::2], heads, dh ** -0.5)
    sp = lambda z: z.reshape(z.shape[0], -1, heads, dh).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", sp(xf[::2, :, :c]), sp(xf[1::2, :, c:])) * dh ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), sp(vf[1::2])).permute(0, 2, 1, 3).reshape(2, t, c)
    assert rel(y.float(), ref) < TOL[dtype]
    # the same with V row-major inside a stacked k|v tensor of the partner half (the production call)
    kvd = torch.cat([xd[:, :, c:], vd], dim=2)
    y2 = ops.attention(xd[::2, :, :c], kvd[1::2, :, :c], kvd[1::2, :, c:], heads, dh ** -0.5, v_rows=True)
    assert rel(y2.float(), ref) < TOL[dtype]
    tune.setenv("MOBI_ATTN_NW", "8")                         # the same through 8-wave blocks, both schedules
    y3 = ops.attention(xd[::2, :, :c], kvd[1::2, :, :c], kvd[1::2, :, c:], heads, dh ** -0.5, v_rows=True)
    tune.setenv("MOBI_ATTN_SP", "1")
    y4 = ops.attention(xd[::2, :, :c], kvd[1::2, :, :c], kvd[1::2, :, c:], heads, dh ** -0.5, v_rows=True)
    assert rel(y3.float(), ref) < TOL[dtype]
    assert rel(y4.float(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tk", [1, 2, 8])
def test_ctx_attention(ops, dtype, tk):
    n, t, heads, dh = 3, 70, 8, 40
    c = heads * dh
    qf, qd = rnd("c.q", (n, t, c), dtype)
    k = W.synth_input(f"c.k{tk}", (n, tk, c))
    v = W.synth_input(f"c.v{tk}", (n, tk, c))
    y = ops.ctx_attention(qd, k.cuda(), v.cuda(), heads, dh ** -0.5)
    sp = lambda z: z.reshape(n, -1, heads, dh).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", sp(qf), sp(k)) * dh ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(n, t, c)
    assert rel(y.float(), ref) < TOL[dtype]


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
def test_skinny_linear_and_timestep_embedding(ops, dtype, tune):
    from mobi_amd._lib import ACT_SILU
    from mobi_amd.ldm.modules.diffusionmodules.util import timestep_embedding
    from oracle.unet import timestep_embedding as ref_temb
    x = W.synth_input("s.x", (5, 2, 320))
    wf = torch.from_numpy(W.synth_param("s.weight", (1288, 320))).to(dtype).float()
    b = torch.from_numpy(W.synth_param("s.bias", (1288,)))
    y = ops.skinny_linear(x.cuda()[:, 0], wf.to(dtype).cuda(), b.cuda(), pre_act=ACT_SILU, post_act=ACT_SILU)
    ref = F.silu(F.linear(F.silu(x[:, 0]), wf, b))
    assert rel(y, ref) < 2e-5
    # the matrix-core forms (k % 32 == 0: 16 columns per block; n >= 8192: 64 per block) and the vector-ALU kernel (other k,
    # MOBI_SKINNY_MFMA=0) on ragged column counts, 1 / 3 / 16 rows, GELU behind, k beyond one 512-deep chunk
    for m, k, n, post in ((16, 1280, 8203, 0), (3, 1280, 1283, 2), (1, 352, 77, 0), (16, 328, 40, 0)):
        xs = W.synth_input(f"s2.x{m}.{k}", (m, k))
        ws = torch.from_numpy(W.synth_param(f"s2.w{k}.{n}", (n, k))).to(dtype)
        bs = torch.from_numpy(W.synth_param(f"s2.b{n}", (n,)))
        ref2 = F.linear(xs.double(), ws.double(), bs.double())
        ref2 = F.gelu(ref2) if post == 2 else ref2
        for env in (None, "0"):
            if env is not None:
                tune.setenv("MOBI_SKINNY_MFMA", env)
            y2 = ops.skinny_linear(xs.cuda(), ws.cuda(), bs.cuda(), post_act=post)
            assert y2.shape == (m, n) and rel(y2.double(), ref2) < 2e-6, (m, k, n, env)
        tune.delenv("MOBI_SKINNY_MFMA")
    t = torch.tensor([1, 21, 500, 981, 999], dtype=torch.long)
    e = timestep_embedding(t.cuda(), 320)
    assert rel(e, ref_temb(t, 320)) < 1e-6


@pytest.mark.parametrize("dtype", DT)
def test_small_convs(ops, dtype):
    # 9-channel input conv from three fp32 NCHW sources (ddim.py:170 + input_blocks.0)
    a, b, m = W.synth_input("sc.a", (2, 4, 12, 12)), W.synth_input("sc.b", (2, 4, 12, 12)), \
        (W.synth_input("sc.m", (2, 1, 12, 12)) > 0).float()
    w = torch.from_numpy(W.synth_param("sc.weight", (64, 9, 3, 3)))
    bias = torch.from_numpy(W.synth_param("sc.bias", (64,)))
    y = ops.conv_small_cin([a.cuda(), b.cuda(), m.cuda()], w.reshape(64, -1).cuda(), bias.cuda(), 3, 3, (1, 1), dtype)
    ref = F.conv2d(torch.cat([a, b, m], 1), w, bias, padding=1).permute(0, 2, 3, 1)
    assert rel(y.float(), ref) < TOL[dtype]
    # 1x5 lidar conv_in, fp32 NCHW out
    r = W.synth_input("sc.r", (2, 2, 6, 20))
    w15 = torch.from_numpy(W.synth_param("sc15.weight", (5, 2, 1, 5)))
    y = ops.conv_small_cin([r.cuda()], w15.reshape(5, -1).cuda(), None, 1, 5, (0, 2), dtype, out_f32_nchw=True)
    assert rel(y, F.conv2d(r, w15, None, padding=(0, 2))) < 2e-6
    # few output channels + clamp
    xf, xd = rnd("sc.x", (2, 9, 7, 64), dtype)
    wo = torch.from_numpy(W.synth_param("sco.weight", (3, 64, 3, 3))).to(dtype).float()
    bo = torch.from_numpy(W.synth_param("sco.bias", (3,)))
    y = ops.conv_small_cout(xd, ops.pack_conv(wo, bo, dtype, "cuda"), clamp=(-0.5, 0.5))
    ref = F.conv2d(xf.permute(0, 3, 1, 2), wo, bo, padding=1).clamp(-0.5, 0.5)
    assert y.shape == ref.shape and rel(y, ref) < 1e-4
    wl = torch.from_numpy(W.synth_param("scl.weight", (2, 64, 1, 5))).to(dtype).float()
    y = ops.conv_small_cout(xd, ops.pack_conv(wl, None, dtype, "cuda"))
    assert rel(y, F.conv2d(xf.permute(0, 3, 1, 2), wl, None, padding=(0, 2))) < 1e-4


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cin,cout,h,w,kh,kw,clamp", [(320, 4, 9, 32, 3, 3, None), (128, 3, 5, 48, 3, 3, (-1.0, 1.0)),
                                                     (128, 2, 6, 16, 1, 5, None), (320, 8, 3, 16, 3, 3, None)])
def test_conv_small_cout_matrix_core_form(ops, dtype, cin, cout, h, w, kh, kw, clamp, tune):
    """The few-output-channel convolution on the matrix cores (cin 320 / 128, w % 16 == 0: the UNet's and the VAE decoders'
    output convolutions) against fp32 torch and against the one-wave-per-pixel kernel (MOBI_COUT_MFMA=0); borders on all
    four sides, the 1 x 5 lidar tap shape, clamp, 2 / 3 / 4 / 8 output channels."""
    name = f"csm.{cin}.{cout}.{h}.{w}.{kh}{kw}"
    xf, xd = rnd(name + ".x", (3, h, w, cin), dtype)
    wt = torch.from_numpy(W.synth_param(name + ".weight", (cout, cin, kh, kw))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param(name + ".bias", (cout,)))
    pw = ops.pack_conv(wt, bias, dtype, "cuda")
    pad = (kh // 2, kw // 2)
    y = ops.conv_small_cout(xd, pw, pad=pad, clamp=clamp)
    ref = F.conv2d(xf.permute(0, 3, 1, 2), wt, bias, padding=pad)
    if clamp:
        ref = ref.clamp(*clamp)
    assert y.shape == ref.shape and y.dtype == torch.float32 and rel(y, ref) < 2e-5 * (1 if dtype == torch.float16 else 1)
    tune.setenv("MOBI_COUT_MFMA", "0")
    y0 = ops.conv_small_cout(xd, pw, pad=pad, clamp=clamp)
    assert rel(y, y0) < 2e-5


def test_layout_and_index_ops_bit_exact(ops):
    x = W.synth_input("l.x", (2, 40, 5, 7))
    for dtype in DT:
        y = ops.to_nhwc(x.cuda(), dtype)
        assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).to(dtype))
        assert torch.equal(ops.to_nchw_f32(y).cpu(), x.to(dtype).float())
    m = (W.synth_input("l.m", (3, 1, 64, 64)) > 0).float()
    for size in (8, 16, 24):
        assert torch.equal(ops.nearest_resize(m.cuda(), size, size).cpu(), F.interpolate(m, size=size, mode="nearest"))
    z = torch.zeros(3, 9, 8, 8).cuda()
    ops.nearest_resize(m.cuda(), 8, 8, out=z, c_off=8)
    assert torch.equal(z[:, 8].cpu(), F.interpolate(m, size=8, mode="nearest")[:, 0]) and float(z[:, :8].abs().sum()) == 0


def test_sampler_arithmetic_bit_exact(ops):
    """fp32 latent update, CFG mix, PLMS mixes, mask compositing: the kernels are compiled without
    FMA contraction and follow the reference's operation order -> identical bits to torch CPU."""
    from oracle import sampler as S
    sch = S.Schedule(50, eta=1.0)
    x, e, eu, nz = (W.synth_input("sa." + k, (4, 4, 16, 16)) for k in "xeun")
    index = 17
    ref_e = eu + 5.0 * (e - eu)
    ref_prev, ref_pred = S._x_prev(sch, index, x, ref_e, nz)
    xp, pr, eo = ops.ddim_step(x.cuda(), e.cuda(), e_uncond=eu.cuda(), noise=nz.cuda(), cfg_scale=5.0,
                               a_t=float(sch.alphas[index]), a_prev=float(sch.alphas_prev[index]),
                               sigma_t=float(sch.sigmas[index]),
                               sqrt_one_minus_at=float(sch.sqrt_one_minus_alphas[index]), want_e=True)
    assert torch.equal(eo.cpu(), ref_e) and torch.equal(pr.cpu(), ref_pred) and torch.equal(xp.cpu(), ref_prev)
    # mask compositing (ddim.py:145-148)
    x0, mn = W.synth_input("sa.x0", (4, 4, 16, 16)), W.synth_input("sa.mn", (4, 4, 16, 16))
    mask = (W.synth_input("sa.mask", (4, 1, 16, 16)) > 0).float()
    ts = torch.full((4,), 341, dtype=torch.long)
    ref = S.q_sample(sch.buffers, x0, ts, mn) * mask + (1.0 - mask) * x
    got = ops.mask_blend_(x.clone().cuda(), x0.cuda(), mn.cuda(), mask.cuda(),
                          float(sch.buffers["sqrt_alphas_cumprod"][341]),
                          float(sch.buffers["sqrt_one_minus_alphas_cumprod"][341]))
    assert torch.equal(got.cpu(), ref)
    # posterior sample (distributions.py:25-37) into a channel slice
    from oracle.vae import posterior_sample
    mom = W.synth_input("sa.mom", (2, 8, 6, 6)) * 3.0
    n = W.synth_input("sa.n", (2, 4, 6, 6))
    out = torch.zeros(2, 9, 6, 6).cuda()
    ops.posterior_sample(mom.cuda(), n.cuda(), out, 4, 0.18215)
    assert rel(out[:, 4:8].cpu(), 0.18215 * posterior_sample(mom, n)) < 1e-6 and float(out[:, :4].abs().sum()) == 0
    y = ops.lincomb4([x.cuda(), e.cuda(), eu.cuda()], [23 / 12, -16 / 12, 5 / 12])
    assert rel(y.cpu(), (23 * x - 16 * e + 5 * eu) / 12) < 1e-6


def test_error_codes_on_device(ops):
    from mobi_amd import _lib
    x = torch.zeros(1, 4, 4, 48, dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.EngineError):
        ops.groupnorm(x, torch.ones(48).cuda(), torch.zeros(48).cuda(), 1e-5, True)       # C % 32 != 0
    q = torch.zeros(1, 4, 24, dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.EngineError):
        ops.attention(q, q, q.permute(0, 2, 1).contiguous(), 2, 1.0)                       # dh = 12


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("split", [None, 2, 5, 8])
def test_igemm_split_k(ops, dtype, split):
    from mobi_amd import _lib as _l
    if not _l.load().mobi_build_info() & 1:               # shipped build: chunk-major weights are refused, loudly
        x_ = torch.zeros(1, 8, 8, 64, dtype=dtype, device="cuda")
        pw_ = ops.pack_conv(torch.zeros(64, 64, 3, 3), None, dtype, "cuda", chunk_major=True)
        with pytest.raises(_l.EngineError):
            ops.igemm(x_, pw_)
    """Small-m / long-k convolution (the 8x8 UNet level): k cut over workgroups, fp32 slabs, reduce launch
    with bias + per-image vector + residual.  split=None exercises the library's own plan."""
    xf, xd = rnd("sk.x", (2, 8, 8, 640), dtype)
    x1f, x1d = rnd("sk.x1", (2, 8, 8, 320), dtype)
    rf, rd = rnd("sk.res", (2, 8, 8, 320), dtype)
    wf = torch.from_numpy(W.synth_param("sk.weight", (320, 960, 3, 3))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param("sk.bias", (320,)))
    rv = W.synth_input("sk.rowvec", (2, 320))
    ref = _conv_ref(torch.cat([xf, x1f], 3), wf, bias) + rv[:, None, None, :] + rf
    for chunk_major in _k_orders():
        pw = ops.pack_conv(wf, bias, dtype, "cuda", chunk_major=chunk_major)
        y = ops.igemm(xd, pw, x2=x1d, rowvec=rv.cuda(), residual=rd, split_k=split)
        assert rel(y.float(), ref) < TOL[dtype], chunk_major
    if split is None:
        from mobi_amd import _lib
        import ctypes as C
        p = _lib.IgemmParams()
        p.batch, p.hout, p.wout, p.n_packed, p.kh, p.kw, p.c0, p.groups = 2, 8, 8, 320, 3, 3, 960, 1
        assert _lib.load().mobi_igemm_plan_splits(C.byref(p)) > 1


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [
    # n, side, cin, cout, taps, split, c1 of the GroupNorm's second source, operands of the producer's epilogue, keep
    dict(n=16, side=8, cin=640, cout=1280, k=3, split=8, c1=0, rowvec=True, resid=False, keep=False),     # 8 x 8 level, conv1 -> out_layers
    dict(n=16, side=8, cin=640, cout=1280, k=3, split=8, c1=1280, rowvec=False, resid=True, keep=True),   # ... -> next block's concat
    dict(n=16, side=16, cin=320, cout=1280, k=3, split=4, c1=0, rowvec=False, resid=True, keep=True),     # 16 x 16 level
    dict(n=8, side=32, cin=320, cout=320, k=3, split=4, c1=0, rowvec=True, resid=False, keep=False),      # 8-byte pieces (mobi_nusc_256's top level)
    dict(n=8, side=4, cin=640, cout=1280, k=3, split=16, c1=1280, rowvec=False, resid=False, keep=True),  # 4 x 4 level, 16 slabs
    dict(n=4, side=8, cin=640, cout=640, k=1, split=3, c1=320, rowvec=False, resid=True, keep=True),      # ragged slab count, 1 x 1
    dict(n=3, side=5, cin=320, cout=640, k=3, split=5, c1=0, rowvec=True, resid=True, keep=True),         # ragged rows, every operand
], ids=lambda c: f"n{c['n']}s{c['side']}c{c['cout']}+{c['c1']}x{c['split']}")
def test_groupnorm_sums_split_k_slabs(ops, dtype, case, tune):
    """mobi_split_source: a split-K launch with defer_finish leaves its fp32 slabs, the GroupNorm that consumes the result
    sums them while it loads (ascending from zero, + bias, + per-image vector, + residual, one rounding): BIT FOR BIT the
    reduce launch followed by the plain GroupNorm, in the output, in the `finished` tensor it writes for the other readers,
    and through mobi_igemm_finish; and against fp32 torch like every GroupNorm."""
    from mobi_amd import _lib
    n, side, cin, cout, k, c1 = case["n"], case["side"], case["cin"], case["cout"], case["k"], case["c1"]
    tag = f"ss{n}.{side}.{cin}.{cout}.{k}"
    xf, xd = rnd(tag + ".x", (n, side, side, cin), dtype)
    wf = (torch.from_numpy(W.synth_param(tag + ".weight", (cout, cin, k, k))) * 2.0).to(dtype).float()
    bias = torch.from_numpy(W.synth_param(tag + ".bias", (cout,)))
    rv = W.synth_input(tag + ".rowvec", (n, cout)).cuda() if case["rowvec"] else None
    rf, rd = rnd(tag + ".res", (n, side, side, cout), dtype) if case["resid"] else (None, None)
    x2f, x2d = rnd(tag + ".x2", (n, side, side, c1), dtype) if c1 else (None, None)
    C = cout + c1
    g = torch.from_numpy(W.synth_param(tag + ".g.weight", (C,))).cuda()
    b = torch.from_numpy(W.synth_param(tag + ".g.bias", (C,))).cuda()
    pw = ops.pack_conv(wf, bias, dtype, "cuda")
    if cout + c1 == 320:
        # the 8-byte-piece geometry (C = 320) takes slabs only on request: measured slower than reduce launch + GroupNorm
        assert _lib.load().mobi_groupnorm_takes_split(cout, c1, n, side * side) == 0
        tune.setenv("MOBI_GN_SPLIT_PW4", "1")
    assert _lib.load().mobi_groupnorm_takes_split(cout, c1, n, side * side) == 1

    # the two launches + the plain GroupNorm
    y0 = ops.igemm(xd, pw, rowvec=rv, residual=rd, split_k=case["split"])
    o0 = ops.groupnorm(y0, g, b, 1e-5, True, x2=x2d)
    # deferred: no reduce launch, the GroupNorm sums the slabs
    d = ops.igemm(xd, pw, rowvec=rv, residual=rd, split_k=case["split"], defer="keep" if case["keep"] else "drop")
    assert isinstance(d, ops.Deferred) and 2 <= d.count <= case["split"]
    if case["keep"]:
        d.tensor.fill_(float("nan"))                         # (whatever the GroupNorm does not write would show)
    o1 = ops.groupnorm(d, g, b, 1e-5, True, x2=x2d)
    assert d.done and torch.equal(o1, o0)
    if case["keep"]:
        assert torch.equal(ops.finished(d), y0)
    # ... or mobi_igemm_finish does, for a reader that is not a GroupNorm
    d2 = ops.igemm(xd, pw, rowvec=rv, residual=rd, split_k=case["split"], defer="keep")
    assert torch.equal(ops.finished(d2), y0) and d2.done
    # and a Deferred that reaches igemm as a source / residual is finished on the way
    d3 = ops.igemm(xd, pw, rowvec=rv, residual=rd, split_k=case["split"], defer="keep")
    pw1 = ops.pack_conv(torch.eye(cout).view(cout, cout, 1, 1), None, dtype, "cuda")
    assert torch.equal(ops.igemm(d3, pw1), ops.igemm(y0, pw1))

    ref = _conv_ref(xf, wf, bias, pad=(k // 2, k // 2))
    if rv is not None:
        ref = ref + rv.cpu()[:, None, None, :]
    if rf is not None:
        ref = ref + rf
    assert rel(y0.float(), ref) < TOL[dtype]
    ref_in = y0.float().cpu() if x2f is None else torch.cat([y0.float().cpu(), x2f], dim=3)
    gref = F.silu(F.group_norm(ref_in.permute(0, 3, 1, 2), 32, g.cpu(), b.cpu(), 1e-5))
    assert rel(o1.float().permute(0, 3, 1, 2), gref) < TOL[dtype]


def test_split_source_argument_checks(ops):
    """What mobi_groupnorm refuses of a split source, and what mobi_igemm refuses of defer_finish (no launch happens)."""
    import ctypes as C
    from mobi_amd import _lib
    lib = _lib.load()
    dtype = torch.float16
    xd = torch.zeros(2, 8, 8, 640, dtype=dtype, device="cuda")
    pw = ops.pack_conv(torch.zeros(640, 640, 3, 3), None, dtype, "cuda")
    assert isinstance(ops.igemm(xd, pw, split_k=1, defer="keep"), torch.Tensor)            # no split: a tensor as always
    d = ops.igemm(xd, pw, split_k=4, defer="keep")
    q = d.params
    assert lib.mobi_igemm_slab_count(C.byref(q)) == d.count == 4
    q.out_mode = 2                                                                          # fp32 rows cannot be deferred
    assert lib.mobi_igemm(C.byref(q), None) == -2
    q.out_mode = 0
    g = torch.ones(640, device="cuda")
    ws = torch.empty(lib.mobi_groupnorm_workspace_bytes(2, 64), dtype=torch.uint8, device="cuda")
    out = torch.empty(2, 8, 8, 640, dtype=dtype, device="cuda")
    ss = _lib.SplitSource()
    ss.slabs, ss.count, ss.row_stride = q.ws, 4, 640
    p = _lib.GroupNormParams()
    p.c0, p.batch, p.hw, p.gamma, p.beta, p.eps, p.silu = 640, 2, 64, g.data_ptr(), g.data_ptr(), 1e-5, 1
    p.out, p.ws, p.dtype, p.src0_split = out.data_ptr(), ws.data_ptr(), 0, C.pointer(ss)
    assert lib.mobi_groupnorm(C.byref(p), None) == 0
    ss.count = 1
    assert lib.mobi_groupnorm(C.byref(p), None) == -1                                       # fewer than two slabs
    ss.count, ss.row_stride = 4, 320
    assert lib.mobi_groupnorm(C.byref(p), None) == -1                                       # rows shorter than the tensor's
    ss.row_stride, p.out_mode = 640, 2
    assert lib.mobi_groupnorm(C.byref(p), None) == -2                                       # precise outputs: the two-launch form
    p.out_mode, p.hw = 0, 64 * 64 * 4
    assert lib.mobi_groupnorm_takes_split(640, 0, 2, 64 * 64 * 4) == 0
    assert lib.mobi_groupnorm(C.byref(p), None) == -2                                       # too large for the register form
    d.finish()
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("wm", ["2", "4"])
def test_igemm_block_heights(ops, dtype, wm, tune):
    """Both block shapes of the implicit-GEMM kernel (4 waves x 128 pixels, 8 waves x 256 pixels) on the same
    convolutions, incl. ragged pixel counts, two sources, GEGLU and the transposed epilogue."""
    from mobi_amd._lib import OUT_TRANSPOSED
    tune.setenv("MOBI_IGEMM_WM", wm)
    for name, cin, cout, kh, kw, h, w, stride, up, asym in IGEMM_CASES[:7]:
        xf, xd = rnd("x." + name, (3, h, w, cin), dtype)
        wf = torch.from_numpy(W.synth_param(name + ".weight", (cout, cin, kh, kw))).to(dtype).float()
        bias = torch.from_numpy(W.synth_param(name + ".bias", (cout,)))
        pad = (kh // 2, kw // 2)
        for chunk_major in _k_orders():              # both k orders of the packed weights (development build)
            pw = ops.pack_conv(wf, bias, dtype, "cuda", chunk_major=chunk_major)
            if asym:
                y = ops.igemm(xd, pw, stride=2, pad=(0, 0), hout=(h + 1 - 3) // 2 + 1, wout=(w + 1 - 3) // 2 + 1)
            else:
                y = ops.igemm(xd, pw, stride=stride, pad=pad, upsample=up)
            assert rel(y.float(), _conv_ref(xf, wf, bias, stride, pad, up, asym)) < TOL[dtype], (name, chunk_major)
    xf, xd = rnd("bh.x", (2, 300, 320), dtype)
    wf = torch.from_numpy(W.synth_param("bh.g.weight", (2560, 320))).to(dtype).float()
    bf = torch.from_numpy(W.synth_param("bh.g.bias", (2560,)))
    y = ops.linear(xd, ops.pack_geglu(wf, bf, dtype, "cuda"))
    a, gate = F.linear(xf, wf, bf).chunk(2, dim=-1)
    assert rel(y.float(), a * F.gelu(gate)) < TOL[dtype]
    wv = torch.from_numpy(W.synth_param("bh.v.weight", (320, 320))).to(dtype).float()
    xs_f, xs_d = rnd("bh.xs", (2, 304, 320), dtype)
    yt = ops.linear(xs_d, ops.pack_linear(wv, None, dtype, "cuda"), out_mode=OUT_TRANSPOSED)
    assert rel(yt.float(), F.linear(xs_f, wv).permute(0, 2, 1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("blocks", [None, "3"])
def test_igemm_persistent_register_epilogue(ops, dtype, blocks, tune):
    """The persistent direct-to-LDS kernel on full 256-pixel tiles: register epilogue (lane-row exchange, counted
    vector-memory waits) vs the LDS-staged epilogue of the same kernel, both against the fp32 reference.
    1x1 + bias + residual (80- and 64-column wave tiles), 3x3 over two sources, GEGLU.
    blocks="3": three persistent blocks walk all output tiles, so the k-tile sequence rolls across many tiles."""
    tune.setenv("MOBI_IGEMM_WM", "4")
    if blocks:
        tune.setenv("MOBI_IGEMM_PERSIST_BLOCKS", blocks)
    xf, xd = rnd("pd.x", (4, 32, 32, 320), dtype)
    rf, rd = rnd("pd.r", (4, 32, 32, 320), dtype)
    r2f, r2d = rnd("pd.r2", (4, 32, 32, 128), dtype)
    x1f, x1d = rnd("pd.x1", (4, 32, 32, 64), dtype)
    w1 = torch.from_numpy(W.synth_param("pd.w1", (320, 320, 1, 1))).to(dtype).float()
    b1 = torch.from_numpy(W.synth_param("pd.b1", (320,)))
    w2 = torch.from_numpy(W.synth_param("pd.w2", (128, 320, 1, 1))).to(dtype).float()
    w3 = torch.from_numpy(W.synth_param("pd.w3", (320, 384, 3, 3))).to(dtype).float()
    b3 = torch.from_numpy(W.synth_param("pd.b3", (320,)))
    wg = torch.from_numpy(W.synth_param("pd.wg", (2560, 320))).to(dtype).float()
    bg = torch.from_numpy(W.synth_param("pd.bg", (2560,)))
    ref1 = _conv_ref(xf, w1, b1, pad=(0, 0)) * 0.5 + 0.5 * b1 + rf          # scale applies before the bias
    ref2 = _conv_ref(xf, w2, None, pad=(0, 0)) + r2f
    ref3 = _conv_ref(torch.cat([xf, x1f], 3), w3, b3)
    a, gate = F.linear(xf.reshape(4, 1024, 320), wg, bg).chunk(2, dim=-1)
    refg = a * F.gelu(gate)
    p1, p2 = ops.pack_conv(w1, b1, dtype, "cuda"), ops.pack_conv(w2, None, dtype, "cuda")
    p3, pg = ops.pack_conv(w3, b3, dtype, "cuda"), ops.pack_geglu(wg, bg, dtype, "cuda")
    outs = {}
    for direct in ("1", "0"):
        tune.setenv("MOBI_IGEMM_EPI_DIRECT", direct)
        y1 = ops.igemm(xd, p1, residual=rd, scale=0.5)
        y2 = ops.igemm(xd, p2, residual=r2d)
        y3 = ops.igemm(xd, p3, x2=x1d)
        yg = ops.linear(xd.view(4, 1024, 320), pg)
        assert rel(y1.float(), ref1) < TOL[dtype], direct
        assert rel(y2.float(), ref2) < TOL[dtype], direct
        assert rel(y3.float(), ref3) < TOL[dtype], direct
        assert rel(yg.float(), refg) < TOL[dtype], direct
        outs[direct] = (y1, y2, y3, yg)
    for ya, yb in zip(outs["1"], outs["0"]):                      # same fp32 arithmetic, one rounding: near-identical
        assert rel(ya.float(), yb.float()) < 2e-3


# (images, h, w, cin, cin2, cout, k, residual, rowvec, geglu, persistent blocks)
PP_CASES = [
    (3, 16, 16, 64, 0, 160, 3, False, False, False, 0),       # one k-tile per tap, padding on every side, one tile / block
    (3, 16, 16, 192, 0, 160, 1, False, False, False, 0),      # three k-tiles: the shortest loop the kernel accepts
    (4, 32, 32, 320, 0, 320, 3, True, False, False, 3),       # few blocks walk many tiles (deferred epilogue, ragged)
    (4, 32, 32, 128, 0, 128, 3, False, True, False, 5),       # 64-wide wave tiles, per-image vector, no residual
    (8, 8, 8, 128, 0, 320, 3, True, True, False, 2),          # 8x8 images: four images per 256-pixel tile
    (2, 64, 64, 320, 0, 640, 1, True, True, False, 7),        # 1x1, residual + per-image vector
    (2, 32, 32, 128, 64, 160, 3, True, False, False, 3),      # two sources (the un-materialised concat)
    (1, 128, 128, 128, 0, 128, 3, True, False, False, 0),     # rows wider than a wave's 64 pixels
    (2, 16, 512, 64, 0, 128, 3, False, False, False, 0),      # half a row per tile
    (2, 64, 64, 320, 0, 640, 1, False, False, True, 6),       # GEGLU register epilogue
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", PP_CASES, ids=[f"pp{i}" for i in range(len(PP_CASES))])
def test_igemm_pingpong(ops, dtype, case, tune):
    """The ping-pong direct-to-LDS kernel against a torch fp32 convolution: the 256-pixel geometry is forced on small problems, a handful of persistent blocks walk many
    output tiles, and the library must report that it runs the ping-pong variant."""
    import ctypes as C
    from mobi_amd import _lib
    n, h, w, cin, cin2, cout, k, res, rowvec, geglu, blocks = case
    tune.setenv("MOBI_IGEMM_WM", "4")
    if blocks:
        tune.setenv("MOBI_IGEMM_PERSIST_BLOCKS", str(blocks))
    name = "pp." + ".".join(str(int(v)) for v in case)
    xf, xd = rnd(name + ".x", (n, h, w, cin), dtype)
    x2f, x2d = rnd(name + ".x2", (n, h, w, cin2), dtype) if cin2 else (None, None)
    ctot = cin + cin2
    wf = torch.from_numpy(W.synth_param(name + ".weight", ((2 if geglu else 1) * cout, ctot, k, k))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param(name + ".bias", ((2 if geglu else 1) * cout,)))
    rf, rd = rnd(name + ".res", (n, h, w, cout), dtype) if res else (None, None)
    rv = W.synth_input(name + ".rv", (n, cout)) if rowvec else None
    xin = xf if x2f is None else torch.cat([xf, x2f], 3)
    ref = _conv_ref(xin, wf, None if rowvec else bias, 1, (k // 2, k // 2))
    if geglu:
        a_, g_ = ref.chunk(2, dim=-1)
        ref = a_ * F.gelu(g_)
    if rv is not None:
        ref = ref + rv[:, None, None, :]
    if rf is not None:
        ref = ref + rf
    for halo in ("0",):
        if geglu:
            pw = ops.pack_geglu(wf[:, :, 0, 0], bias, dtype, "cuda")
            y = ops.linear(xd.view(n, h * w, cin), pw).view(n, h, w, cout)
        else:
            pw = ops.pack_conv(wf, None if rowvec else bias, dtype, "cuda")
            y = ops.igemm(xd, pw, x2=x2d, residual=rd, rowvec=None if rv is None else rv.cuda(), rowvec_has_bias=rowvec)
        assert torch.isfinite(y.float()).all(), (case, halo)
        assert rel(y.float(), ref) < TOL[dtype], (case, halo)
    # the launch really is the ping-pong variant
    p = _lib.IgemmParams()
    p.src0, p.weight, p.out = 256, 256, 256                   # non-null, 16-byte aligned placeholders (no launch)
    p.c0, p.c1, p.batch, p.hin, p.win, p.hout, p.wout = cin, cin2, n, h, w, h, w
    if cin2:
        p.src1 = 256
    p.kh = p.kw = k
    p.stride, p.pad_h, p.pad_w, p.groups = 1, k // 2, k // 2, 1
    p.cout, p.n_packed = cout, (2 if geglu else 1) * cout
    p.epilogue = _lib.EPI_GEGLU if geglu else _lib.EPI_NONE
    p.scale, p.dtype = 1.0, _lib.MOBI_F16 if dtype == torch.float16 else _lib.MOBI_BF16
    assert _lib.load().mobi_igemm_kernel_variant(C.byref(p)) == 3


# (images, h, w, cin, cin2, cout, k, residual, rowvec, geglu, split)
SM_CASES = [
    (2, 16, 16, 640, 0, 320, 3, True, True, False, None),     # 512 pixels: four full 128-pixel tiles, residual + per-image vector
    (8, 8, 8, 320, 0, 640, 1, True, False, False, None),      # 8 x 8 images: two images per tile, bias + residual
    (2, 16, 16, 256, 64, 320, 3, False, False, False, None),  # two sources
    (2, 16, 16, 320, 0, 320, 1, False, False, True, None),    # GEGLU (640 packed columns)
    (2, 8, 8, 1280, 0, 320, 3, True, True, False, 5),         # split-K: fp32 slabs straight from the accumulators
    (4, 8, 8, 640, 0, 1280, 1, True, False, False, 2),        # split-K, 1x1, 256-wide... 1280 = 8 x 160 column tiles
    (1, 16, 8, 128, 0, 128, 3, True, False, False, None),     # 128-wide tiles (four 16-column MFMA tiles per wave)
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", SM_CASES, ids=[f"sm{i}" for i in range(len(SM_CASES))])
def test_igemm_ring128_epilogues(ops, dtype, case, tune):
    """The 128 x 160 (128) ring tiles (small m) with the register epilogue / register slab stores and with the LDS-staged
    epilogue: against a torch fp32 convolution, and bit for bit against each other (the same sums, the same rounding)."""
    n, h, w, cin, cin2, cout, k, res, rowvec, geglu, split = case
    name = "sm." + ".".join(str(v) for v in case)
    xf, xd = rnd(name + ".x", (n, h, w, cin), dtype)
    x2f, x2d = rnd(name + ".x2", (n, h, w, cin2), dtype) if cin2 else (None, None)
    wf = torch.from_numpy(W.synth_param(name + ".weight", ((2 if geglu else 1) * cout, cin + cin2, k, k))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param(name + ".bias", ((2 if geglu else 1) * cout,)))
    rf, rd = rnd(name + ".res", (n, h, w, cout), dtype) if res else (None, None)
    rv = W.synth_input(name + ".rv", (n, cout)) if rowvec else None
    ref = _conv_ref(xf if x2f is None else torch.cat([xf, x2f], 3), wf, None if rowvec else bias, 1, (k // 2, k // 2))
    if geglu:
        a_, g_ = ref.chunk(2, dim=-1)
        ref = a_ * F.gelu(g_)
    if rv is not None:
        ref = ref + rv[:, None, None, :]
    if rf is not None:
        ref = ref + rf
    tune.setenv("MOBI_IGEMM_WM", "2")
    tune.setenv("MOBI_IGEMM_WIDE", "0")
    outs = {}
    for direct in ("1", "0"):
        tune.setenv("MOBI_IGEMM_SM_DIRECT", direct)
        if geglu:
            y = ops.linear(xd.view(n, h * w, cin), ops.pack_geglu(wf[:, :, 0, 0], bias, dtype, "cuda")).view(n, h, w, cout)
        else:
            pw = ops.pack_conv(wf, None if rowvec else bias, dtype, "cuda")
            y = ops.igemm(xd, pw, x2=x2d, residual=rd, rowvec=None if rv is None else rv.cuda(), rowvec_has_bias=rowvec,
                          split_k=split)
        assert torch.isfinite(y.float()).all() and rel(y.float(), ref) < TOL[dtype], (case, direct)
        outs[direct] = y
    if split:
        assert torch.equal(outs["1"], outs["0"])                  # identical slabs, the same reduce launch
    else:
        ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
        d = (outs["1"].float() - outs["0"].float()).abs() / outs["0"].float().abs().clamp_min(1.0)
        assert float(d.max()) <= 2 * ulp


# (images, h, w, cin, cin2, cout, k, residual, rowvec, geglu)
RING_CASES = [
    (2, 64, 64, 320, 0, 320, 1, True, False, False),          # 320-wide tiles (five 16-column MFMA tiles per wave), bias + residual
    (2, 64, 64, 320, 0, 640, 3, True, True, False),           # 3x3, residual + per-image vector (the ResBlock conv)
    (1, 128, 128, 128, 0, 256, 3, True, False, False),        # 256-wide tiles (four MFMA tiles per wave), rows of 128 pixels
    (4, 32, 32, 128, 64, 320, 3, False, False, False),        # two sources, no residual
    (2, 64, 64, 320, 0, 640, 1, False, False, True),          # GEGLU: 640 outputs = 1280 packed columns
    (2, 32, 32, 256, 0, 256, 1, False, False, True),          # GEGLU on 256-wide tiles
    (3, 16, 16, 192, 0, 320, 1, True, False, False),          # 768 pixels: three full 256-pixel tiles, 16 x 16 images
    (2, 128, 128, 96, 0, 960, 1, False, True, False),         # per-image vector, no residual, 3 column tiles x 128 row tiles, 3 k-steps
    (1, 64, 64, 64, 0, 640, 3, False, False, False),          # k-steps that change tap every two steps (64 channels)
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", RING_CASES, ids=[f"ring{i}" for i in range(len(RING_CASES))])
def test_igemm_ring256_epilogues(ops, dtype, case, tune):
    """The 256 x 320 (256) ring tiles with BOTH epilogues -- registers (bias / per-image vector start the sums, residual rows
    double-buffered over four 32-pixel passes, GEGLU) and LDS-staged -- against a torch fp32 convolution and each other
    (same fp32 sums, one rounding: the outputs differ by at most an ulp of the storage type)."""
    import ctypes as C
    from mobi_amd import _lib
    n, h, w, cin, cin2, cout, k, res, rowvec, geglu = case
    name = "ring." + ".".join(str(int(v)) for v in case)
    xf, xd = rnd(name + ".x", (n, h, w, cin), dtype)
    x2f, x2d = rnd(name + ".x2", (n, h, w, cin2), dtype) if cin2 else (None, None)
    wf = torch.from_numpy(W.synth_param(name + ".weight", ((2 if geglu else 1) * cout, cin + cin2, k, k))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param(name + ".bias", ((2 if geglu else 1) * cout,)))
    rf, rd = rnd(name + ".res", (n, h, w, cout), dtype) if res else (None, None)
    rv = W.synth_input(name + ".rv", (n, cout)) if rowvec else None
    ref = _conv_ref(xf if x2f is None else torch.cat([xf, x2f], 3), wf, None if rowvec else bias, 1, (k // 2, k // 2))
    if geglu:
        a_, g_ = ref.chunk(2, dim=-1)
        ref = a_ * F.gelu(g_)
    if rv is not None:
        ref = ref + rv[:, None, None, :]
    if rf is not None:
        ref = ref + rf
    tune.setenv("MOBI_IGEMM_WIDE", "2")
    outs = {}
    for direct in ("1", "0"):
        tune.setenv("MOBI_IGEMM_RING_DIRECT", direct)
        if geglu:
            y = ops.linear(xd.view(n, h * w, cin), ops.pack_geglu(wf[:, :, 0, 0], bias, dtype, "cuda")).view(n, h, w, cout)
        else:
            pw = ops.pack_conv(wf, None if rowvec else bias, dtype, "cuda")
            y = ops.igemm(xd, pw, x2=x2d, residual=rd, rowvec=None if rv is None else rv.cuda(), rowvec_has_bias=rowvec)
        assert torch.isfinite(y.float()).all() and rel(y.float(), ref) < TOL[dtype], (case, direct)
        outs[direct] = y.float()
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    assert float(((outs["1"] - outs["0"]).abs() / outs["0"].abs().clamp_min(1.0)).max()) <= 2 * ulp
    p = _lib.IgemmParams()
    p.src0, p.weight, p.out = 256, 256, 256
    p.c0, p.c1, p.batch, p.hin, p.win, p.hout, p.wout = cin, cin2, n, h, w, h, w
    if cin2:
        p.src1 = 256
    p.kh = p.kw = k
    p.stride, p.pad_h, p.pad_w, p.groups = 1, k // 2, k // 2, 1
    p.cout, p.n_packed = cout, (2 if geglu else 1) * cout
    p.epilogue = _lib.EPI_GEGLU if geglu else _lib.EPI_NONE
    p.scale, p.dtype = 1.0, _lib.MOBI_F16 if dtype == torch.float16 else _lib.MOBI_BF16
    assert _lib.load().mobi_igemm_kernel_variant(C.byref(p)) == 5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,t,c,heads,strided", [(3, 100, 320, 8, False), (2, 64, 640, 8, True), (2, 37, 1280, 8, False),
                                                 (3, 401, 320, 8, False), (2, 2101, 640, 8, True),
                                                 (1, 256, 64, 4, False)])
def test_two_key_adapter(ops, dtype, n, t, c, heads, strided, tune):
    """One-pass bbox adapter kernels (LayerNorm statistics, per-head gate logits, gated per-image vectors: token rows in
    registers at C = 320 / 640, the LDS-tile and vector-ALU kernels elsewhere and under MOBI_TKA_MFMA) against the same
    formula in torch fp32; ragged token counts, three channel widths, a batch-strided view updated in place."""
    name = f"tka.{n}.{t}.{c}.{heads}"
    full = 2 * n if strided else n
    xf, xd = rnd(name + ".x", (full, t, c), dtype, scale=2.0)
    a = W.synth_input(name + ".a", (n, heads, c)) * 0.05
    u = W.synth_input(name + ".u", (n, heads, c))
    b = W.synth_input(name + ".b", (n, c))
    cc = W.synth_input(name + ".c", (n, heads))
    xs = xf[::2] if strided else xf
    mean = xs.mean(-1, keepdim=True)
    rstd = (xs.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    z = rstd * (torch.einsum("ntc,nhc->nth", xs, a) - mean * a.sum(-1)[:, None, :]) + cc[:, None, :]
    ref = xs + b[:, None, :] + torch.einsum("nth,nhc->ntc", torch.sigmoid(z), u)
    view = xd[::2] if strided else xd
    y = ops.two_key_adapter(view, a.cuda(), a.sum(-1).contiguous().cuda(), cc.cuda(), u.cuda(), b.cuda(), 1e-5)
    assert rel(y.float(), ref) < TOL[dtype]
    x0 = xd.clone()
    y2 = ops.two_key_adapter(view, a.cuda(), a.sum(-1).contiguous().cuda(), cc.cuda(), u.cuda(), b.cuda(), 1e-5, out=view)
    assert y2.data_ptr() == view.data_ptr() and torch.equal(y2, y)
    if strided:                                                        # the partner images are untouched
        assert torch.equal(xd[1::2].float().cpu(), xf[1::2])
    for form in ("0", "1"):                                            # the vector-ALU and the LDS-tile kernels on the same shapes
        tune.setenv("MOBI_TKA_MFMA", form)
        y3 = ops.two_key_adapter(x0[::2] if strided else x0, a.cuda(), a.sum(-1).contiguous().cuda(), cc.cuda(), u.cuda(),
                                 b.cuda(), 1e-5)
        assert rel(y3.float(), ref) < TOL[dtype], form


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,t,c", [(4, 300, 320), (2, 2048, 640), (6, 700, 640), (4, 37, 1280), (2, 100, 640), (2, 50, 320),
                                   (2, 33, 64)])
def test_two_key_adapter_layernorm_pair(ops, dtype, n, t, c, tune):
    """The adapter kernels' second result: LayerNorm of the result rows, even images with one (gamma, beta), odd images with
    another, against mobi_layernorm on the stored result (same rounded input: only the summation order differs) and against
    fp32 torch; the first result is unchanged by asking for the second; register kernel (C = 320 / 640 with enough rows) and
    vector-ALU kernel (every other shape, MOBI_TKA_MFMA=0); the A/B-only LDS-tile kernel says it does not."""
    name = f"tkaln.{n}.{t}.{c}"
    xf, xd = rnd(name + ".x", (n, t, c), dtype, scale=2.0)
    a = (W.synth_input(name + ".a", (n, 8, c)) * 0.05).cuda()
    u, b, cc = (W.synth_input(name + k, s).cuda() for k, s in ((".u", (n, 8, c)), (".b", (n, c)), (".c", (n, 8))))
    gb = [(torch.from_numpy(W.synth_param(f"{name}.g{i}", (c,))).cuda(), torch.from_numpy(W.synth_param(f"{name}.b{i}", (c,))).cuda())
          for i in range(2)]
    assert ops.two_key_adapter_fuses_ln(c, n * t)                       # the register kernel or the vector-ALU kernel: both write it
    y = ops.two_key_adapter(xd, a, a.sum(-1).contiguous(), cc, u, b, 1e-5)
    y2, (l0, l1) = ops.two_key_adapter(xd, a, a.sum(-1).contiguous(), cc, u, b, 1e-5, ln_pair=(gb[0], gb[1], 1e-5))
    assert torch.equal(y, y2) and l0.shape == l1.shape == (n // 2, t, c)
    for got, half, (g, bt) in ((l0, y[0::2], gb[0]), (l1, y[1::2], gb[1])):
        ref = F.layer_norm(half.float(), (c,), g, bt, 1e-5)
        assert rel(got.float(), ref) < TOL[dtype]
        sep = ops.layernorm(half.contiguous(), g, bt, 1e-5)
        assert rel(got.float(), sep.float()) < TOL[dtype] / 4 and float((got.float() - sep.float()).abs().max()) < 0.07
    tune.setenv("MOBI_TKA_MFMA", "0")                                   # the vector-ALU kernel on every shape
    assert ops.two_key_adapter_fuses_ln(c, n * t)
    y3, (m0, m1) = ops.two_key_adapter(xd, a, a.sum(-1).contiguous(), cc, u, b, 1e-5, ln_pair=(gb[0], gb[1], 1e-5))
    for got, half, (g, bt) in ((m0, y3[0::2], gb[0]), (m1, y3[1::2], gb[1])):
        assert rel(got.float(), F.layer_norm(half.float(), (c,), g, bt, 1e-5)) < TOL[dtype]
    tune.setenv("MOBI_TKA_MFMA", "1")                                   # the LDS-tile kernel (A/B only) does not write it
    if c % 32 == 0 and c <= 640:
        assert not ops.two_key_adapter_fuses_ln(c, n * t)
        with pytest.raises(Exception):
            ops.two_key_adapter(xd, a, a.sum(-1).contiguous(), cc, u, b, 1e-5, ln_pair=(gb[0], gb[1], 1e-5))


@pytest.mark.parametrize("dtype", DT)
def test_igemm_pingpong_upsample_and_stride(ops, dtype, tune):
    """Nearest-x2 upsampling on the load side and a stride-2 convolution through the ping-pong kernel."""
    tune.setenv("MOBI_IGEMM_WM", "4")
    tune.setenv("MOBI_IGEMM_PERSIST_BLOCKS", "3")
    for name, h, cin, cout, stride, up in (("ppup", 8, 128, 160, 1, True), ("pps2", 32, 64, 128, 2, False)):
        xf, xd = rnd(name + ".x", (4, h, h, cin), dtype)
        wf = torch.from_numpy(W.synth_param(name + ".weight", (cout, cin, 3, 3))).to(dtype).float()
        bias = torch.from_numpy(W.synth_param(name + ".bias", (cout,)))
        y = ops.igemm(xd, ops.pack_conv(wf, bias, dtype, "cuda"), stride=stride, upsample=up)
        assert rel(y.float(), _conv_ref(xf, wf, bias, stride, (1, 1), up)) < TOL[dtype], name


def test_range_denorm_vs_reference_golden(ops):
    """mobi_range_denorm against the golden the REFERENCE's inverse_depth_normalization produced: the depth branch
    bit-exact (same operation order, no FMA contraction), the logarithm within 1e-6 absolute; plus the oracle on a
    512 x 512 batch and the pass-through modes."""
    from oracle import postprocess as opost
    from tests.golden_cases import load
    g = load("postprocess")
    depth, inten = ops.range_denorm(g["sample"].cuda(), g["min_d"].cuda(), g["max_d"].cuda(), alpha=float(g["alpha"]))
    assert torch.equal(depth.cpu(), g["depth"])
    assert (inten.cpu() - g["intensity"]).abs().max() < 1e-6
    x = torch.clamp(W.synth_input("post.big", (8, 2, 512, 512)) * 0.9, -1, 1)
    lo = torch.linspace(-0.9, 0.2, 8)
    hi = lo + torch.linspace(0.1, 0.7, 8)
    d, i = ops.range_denorm(x.cuda(), lo.cuda(), hi.cuda(), alpha=0.6)
    dr, ir = opost.range_denorm(x, lo, hi, alpha=0.6)
    assert torch.equal(d.cpu(), dr) and (i.cpu() - ir).abs().max() < 1e-6
    d, i = ops.range_denorm(x.cuda(), None, None, object_norm=False, int_norm=False)
    assert torch.equal(d.cpu(), x[:, [0]]) and torch.equal(i.cpu(), x[:, [1]])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("blocks", [0, 5])
def test_igemm_pingpong_split_k(ops, dtype, blocks, tune):
    """Split-K on the ping-pong kernel: every (tile, k range) block leaves fp32 partial sums from its accumulators, the
    reduce launch adds bias + per-image vector + residual.  Long k (3x3 over 1280 channels), 16 x 16 images, explicit
    split counts, a few persistent blocks walking several tiles."""
    import ctypes as C
    from mobi_amd import _lib
    if blocks:
        tune.setenv("MOBI_IGEMM_PERSIST_BLOCKS", str(blocks))
    n, h, cin, cout = 16, 16, 1280, 320
    xf, xd = rnd("pps.x", (n, h, h, cin), dtype)
    rf, rd = rnd("pps.res", (n, h, h, cout), dtype)
    wf = torch.from_numpy(W.synth_param("pps.weight", (cout, cin, 3, 3))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param("pps.bias", (cout,)))
    rv = W.synth_input("pps.rowvec", (n, cout))
    ref = _conv_ref(xf, wf, bias) + rv[:, None, None, :] + rf
    pw = ops.pack_conv(wf, bias, dtype, "cuda")
    for split in (8, 11):
        y = ops.igemm(xd, pw, rowvec=rv.cuda(), residual=rd, split_k=split)
        assert rel(y.float(), ref) < TOL[dtype], split
    p = _lib.IgemmParams()
    p.src0, p.weight, p.out, p.ws = 256, 256, 256, 256
    p.c0, p.batch, p.hin, p.win, p.hout, p.wout = cin, n, h, h, h, h
    p.kh = p.kw = 3
    p.stride, p.pad_h, p.pad_w, p.groups, p.split_k = 1, 1, 1, 1, 8
    p.cout, p.n_packed, p.scale = cout, cout, 1.0
    p.dtype = _lib.MOBI_F16 if dtype == torch.float16 else _lib.MOBI_BF16
    assert _lib.load().mobi_igemm_kernel_variant(C.byref(p)) == 3


@pytest.mark.parametrize("m,n,k", [(1, 7, 5), (16, 320, 320), (128, 1280, 1280), (33, 100, 770), (2, 640, 64)])
def test_linear_f32(ops, m, n, k):
    """mobi_linear_f32: the per-run folds of the conditioning tokens (fp32 in, fp32 FMA chains, fp32 out) against fp64."""
    x = W.synth_input(f"lf.x{m}.{k}", (m, k + 3))[:, :k].cuda()          # row stride > k
    w = W.synth_input(f"lf.w{n}.{k}", (n, k)).cuda()
    b = W.synth_input(f"lf.b{n}", (n,)).cuda()
    ref = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    y = ops.linear_f32(x, w, b)
    assert y.dtype == torch.float32 and rel(y, ref) < 2e-6
    assert rel(ops.linear_f32(x, w), ref - b.double().cpu()) < 2e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,hidden,residual", [(128, 1280, True), (65536, 1280, True), (300, 1280, False), (1, 64, True),
                                                  (4096 + 17, 320, True)])
def test_ff_geglu_fused(ops, dtype, rows, hidden, residual):
    """mobi_ff_geglu: (x W1v^T + b1v) * gelu(x W1g^T + b1g) . W2^T + b2 + residual in one launch (C = 320) against fp32
    torch with the hidden activation rounded to the storage type (as both engine paths do), and against the two-launch
    path (GEGLU projection, then the output projection)."""
    c = 320
    xf, xd = rnd(f"ff.x{rows}", (rows, c), dtype)
    rf, rd = rnd(f"ff.r{rows}", (rows, c), dtype)
    w1 = torch.from_numpy(W.synth_param("ff.w1.weight", (2 * hidden, c))) * 2.0
    b1 = torch.from_numpy(W.synth_param("ff.w1.bias", (2 * hidden,)))
    w2 = torch.from_numpy(W.synth_param("ff.w2.weight", (c, hidden))) * 2.0
    b2 = torch.from_numpy(W.synth_param("ff.w2.bias", (c,)))
    pf = ops.pack_ff_geglu(w1, b1, w2, b2, dtype, "cuda")
    y = ops.ff_geglu(xd, pf, residual=rd if residual else None)
    w1r, w2r = w1.to(dtype).float(), w2.to(dtype).float()
    h = F.linear(xf, w1r, b1)
    h = (h[:, :hidden] * F.gelu(h[:, hidden:])).to(dtype).float()
    ref = F.linear(h, w2r, b2) + (rf if residual else 0)
    assert y.shape == (rows, c) and rel(y.float(), ref) < TOL[dtype]
    # the two-launch path on the same inputs
    g = ops.linear(xd.view(1, rows, c), ops.pack_geglu(w1, b1, dtype, "cuda"))
    y2 = ops.linear(g, ops.pack_linear(w2, b2, dtype, "cuda"), residual=rd.view(1, rows, c) if residual else None)
    assert rel(y.float(), y2.view(rows, c).float().cpu()) < TOL[dtype]
    # LayerNorm of x inside the kernel (norm3) against LayerNorm as a launch of its own followed by the same kernel
    gam = torch.from_numpy(W.synth_param("ff.ln.weight", (c,))).cuda()
    bet = torch.from_numpy(W.synth_param("ff.ln.bias", (c,))).cuda()
    xs = (xd.float() * 1.7 + 0.4).to(dtype)
    y3 = ops.ff_geglu(xs, pf, residual=xs if residual else None, ln=(gam, bet, 1e-5))
    y4 = ops.ff_geglu(ops.layernorm(xs.view(1, rows, c), gam, bet, 1e-5).view(rows, c), pf, residual=xs if residual else None)
    assert rel(y3.float(), y4.float()) < TOL[dtype] / 2


# (images, h, w, cin, cin2, cout, residual, rowvec, out f32, scale, strided sources[, kernel size])
SMALL_CASES = [
    (4, 8, 8, 1280, 0, 1280, True, False, False, 1.0, False),     # the 8 x 8 level's to_out (+ residual): four k batches per wave
    (2, 16, 16, 320, 0, 320, True, True, False, 1.0, False),      # one batch per wave, residual + per-image vector
    (3, 7, 9, 640, 0, 704, False, False, False, 1.0, False),      # ragged rows (189), 704 = 11 x 64 columns
    (2, 16, 8, 1920, 0, 640, True, False, False, 1.0, False),     # six batches per wave: the request ring wraps
    (2, 8, 8, 640, 0, 96, False, True, True, 0.5, False),         # fp32 output, scale, 96 columns
    (5, 1, 257, 960, 0, 320, True, False, False, 1.0, True),      # token rows of 257 per image, strided images
    (2, 16, 8, 640, 320, 640, True, False, False, 1.0, False),    # two sources (a ResBlock's skip convolution on a concat)
    (8, 4, 4, 1280, 0, 1280, False, True, False, 1.0, False, 3),  # 3 x 3 at the 4 x 4 level: every tap mask, a per-image vector
    (3, 5, 7, 320, 320, 96, True, False, False, 1.0, False, 3),   # 3 x 3 on a concat, ragged rows (105), odd image sides
    (2, 8, 8, 640, 0, 64, False, False, True, 1.0, False, 3),     # 3 x 3, fp32 output
    (1, 1, 1, 320, 0, 32, True, True, False, 1.0, False),         # ONE row, one tile, one batch per wave
    (1, 1, 33, 1280, 0, 32, False, False, False, 2.0, False),     # 33 rows: a full tile and a one-row tile
    (1, 1, 1, 320, 0, 32, False, False, False, 1.0, False, 3),    # 3 x 3 on a single pixel: only the centre tap is inside
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", ["32", None])
@pytest.mark.parametrize("case", SMALL_CASES, ids=[f"small{i}" for i in range(len(SMALL_CASES))])
def test_igemm_small(ops, dtype, case, tile, tune):
    """The small-problem kernel of the 1 x 1 case (csrc/igemm_small.hip), forced and as routed: against
    fp32 torch, against the LDS-ring kernels (same products, another summation order), and bit for bit on repetition."""
    from mobi_amd import _lib
    n, h, w, cin, cin2, cout, res, rowvec, f32, scale, strided = case[:11]
    ks = case[11] if len(case) > 11 else 1
    name = "small." + ".".join(str(v) for v in case)
    xf, xd = rnd(name + ".x", (n, h, w, cin), dtype)
    if strided:
        big = torch.zeros((n, h * w + 3, 1, cin), device="cuda", dtype=dtype)
        big[:, :h * w] = xd.view(n, h * w, 1, cin)
        xd = big[:, :h * w].view(n, h * w, 1, cin)
        xd = xd.as_strided((n, h, w, cin), (xd.stride(0), w * cin, cin, 1))
    x2f, x2d = rnd(name + ".x2", (n, h, w, cin2), dtype) if cin2 else (None, None)
    wf = torch.from_numpy(W.synth_param(name + ".weight", (cout, cin + cin2, ks, ks))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param(name + ".bias", (cout,)))
    rf, rd = rnd(name + ".res", (n, h, w, cout), dtype) if res else (None, None)
    rv = W.synth_input(name + ".rv", (n, cout)) if rowvec else None
    ref = _conv_ref(xf if x2f is None else torch.cat([xf, x2f], 3), wf, None, 1, (ks // 2, ks // 2)) * scale + bias
    if rv is not None:
        ref = ref + rv[:, None, None, :]
    if rf is not None:
        ref = ref + rf
    pw = ops.pack_conv(wf, bias, dtype, "cuda")
    kw = dict(x2=x2d, residual=rd, rowvec=None if rv is None else rv.cuda(), scale=scale,
              out_mode=ops.OUT_ROWS_F32 if f32 else ops.OUT_ROWS)
    if tile is None:
        tune.delenv("MOBI_IGEMM_SMALL")
        if ks == 3:
            tune.setenv("MOBI_IGEMM_SMALL_CONV_M", "4096")         # the 3 x 3 form is not routed by default (measured slower)
    else:
        tune.setenv("MOBI_IGEMM_SMALL", tile)
    sink = []
    ops.set_profiler(sink)
    y = ops.igemm(xd, pw, **kw)
    ops.set_profiler(None)
    assert "kern=small" in sink[-1][-1], sink[-1]                  # every case is small enough to be routed here by itself
    assert torch.isfinite(y.float()).all() and rel(y.float(), ref) < (2e-6 if f32 else TOL[dtype]), (case, tile)
    assert torch.equal(y, ops.igemm(xd, pw, **kw))
    tune.setenv("MOBI_IGEMM_SMALL", "0")
    z = ops.igemm(xd, pw, **kw)
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    d = (y.float() - z.float()).abs() / z.float().abs().clamp_min(1.0)
    assert float(d.max()) <= (1e-5 if f32 else 2 * ulp), (case, tile)


@pytest.mark.parametrize("dtype", DT)
def test_tile_weights_native_equals_restatement(ops, dtype, monkeypatch):
    """mobi_tile_weights (the ring kernels' 1-KiB request images) against the torch.gather restatement it replaces: bit for bit."""
    for n, k in ((16, 32), (320, 320), (1280, 11520), (48, 2880)):
        w = (W.synth_input(f"tile.{n}.{k}", (n, k))).to(dtype).cuda()
        native = ops.tile_weights(w)
        monkeypatch.setattr(ops, "TILE_WEIGHTS_TORCH", True)
        restated = ops.tile_weights(w)
        monkeypatch.setattr(ops, "TILE_WEIGHTS_TORCH", False)
        assert native.shape == restated.shape == (n // 16, k // 32, 16, 4, 8) and torch.equal(native, restated)
    assert ops.tile_weights(torch.zeros(24, 32, device="cuda", dtype=dtype)) is None


# (images, h, cin, cout, k, split, what)
SELF_FINISH_CASES = [
    (4, 16, 1280, 1280, 1, 2, "ring tiles, 64-deep steps, register slab stores (one round of blocks)"),
    (16, 16, 1280, 1280, 1, 2, "ring tiles, 32-deep steps, two blocks per CU"),
    (16, 16, 1280, 1280, 1, 3, "three splits: the splits of a tile land on different XCDs' L2s"),
    (2, 10, 640, 320, 3, 4, "ragged tiles (200 rows): LDS-staged slab stores"),
    (7, 8, 1280, 1280, 3, 3, "448 rows: 4 x 8 tiles of 128 x 160 (the last one ragged)"),
    (16, 8, 1280, 1280, 3, 4, "ping-pong kernel, slabs from the accumulators, one tile per block"),
    (16, 8, 1280, 1280, 3, 8, "more than four splits: the reduce launch (unchanged)"),
    (16, 16, 1280, 1280, 3, 4, "256 x 320 ring tiles with slabs (the 16 x 16 level's 3 x 3 convolutions)"),
    (16, 16, 640, 1280, 3, None, "the library's own plan"),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mode", ["1", "2"], ids=["device_coherent", "same_xcd"])
@pytest.mark.parametrize("case", SELF_FINISH_CASES, ids=[f"sf{i}" for i in range(len(SELF_FINISH_CASES))])
def test_igemm_split_k_finishes_itself(ops, dtype, case, mode, tune):
    """Split-K finished INSIDE the launch (mobi_igemm_params.sync): the workgroup that arrives last at an output tile sums
    the tile's fp32 slabs in split order and applies the epilogue.  Against the two-launch form (slabs + reduce launch):
    BIT-identical (the same sums in the same order, whoever arrives last); the arrival counters are zero again after every
    launch; repeated launches on recycled workspace memory with OTHER inputs in between never read a stale slab; fp32 output."""
    n, h, cin, cout, k, split, _what = case
    tune.setenv("MOBI_IGEMM_FUSED_SPLIT", mode)           # 1: sc1 slab traffic + device-scope counter; 2: the same-XCD form (default)
    xf, xd = rnd("sf.x", (n, h, h, cin), dtype)
    x2f, x2d = rnd("sf.x2", (n, h, h, cin), dtype, scale=-0.7)
    rf, rd = rnd("sf.res", (n, h, h, cout), dtype)
    wf = torch.from_numpy(W.synth_param("sf.weight", (cout, cin, k, k))).to(dtype).float()
    bias = torch.from_numpy(W.synth_param("sf.bias", (cout,)))
    rv = W.synth_input("sf.rowvec", (n, cout)).cuda()
    pw = ops.pack_conv(wf, bias, dtype, "cuda")
    was = ops.FUSED_SPLIT
    try:
        ops.FUSED_SPLIT = False
        two = [ops.igemm(x_, pw, rowvec=rv, residual=rd, split_k=split).clone() for x_ in (xd, x2d)]
        two32 = ops.igemm(xd, pw, residual=rd, split_k=split, out_mode=ops.OUT_ROWS_F32).clone()
        ref = _conv_ref(xf, wf, bias, pad=(k // 2, k // 2)) + rv.cpu()[:, None, None, :] + rf
        assert rel(two[0].float(), ref) < TOL[dtype]
        ops.FUSED_SPLIT = True
        for rep in range(6):                              # the caching allocator hands the same workspace block out again
            for i, x_ in enumerate((xd, x2d)):
                y = ops.igemm(x_, pw, rowvec=rv, residual=rd, split_k=split)
                assert torch.equal(y, two[i]), (rep, i, float((y.float() - two[i].float()).abs().max()))
        y32 = ops.igemm(xd, pw, residual=rd, split_k=split, out_mode=ops.OUT_ROWS_F32)
        assert y32.dtype == torch.float32 and torch.equal(y32, two32)
    finally:
        ops.FUSED_SPLIT = was
    torch.cuda.synchronize()
    for buf in ops._SYNC.values():
        assert int(buf.abs().sum()) == 0                  # every launch left its counters at zero


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,t,c", [(16, 256, 1280), (8, 1024, 640), (4, 100, 320), (2, 64, 1280)])
def test_igemm_groups_of_images(ops, dtype, n, t, c):
    """mobi_igemm_params.groups = 2: the first half of the images times the first stacked matrix, the second half times the
    second -- the cross-modal to_q of the camera images and of the lidar images as ONE launch (attention.py:245-263 of the
    reference computes them as two Linear calls).  Against fp32 torch; full tiles (register epilogue, request images) and
    ragged ones (LDS-staged epilogue)."""
    xf, xd = rnd("grp.x", (n, t, c), dtype)
    w0 = torch.from_numpy(W.synth_param("grp.w0", (c, c))).to(dtype).float()
    w1 = torch.from_numpy(W.synth_param("grp.w1", (c, c))).to(dtype).float()
    pw = ops.pack_linear(torch.cat([w0, w1], 0), None, dtype, "cuda")
    y = ops.linear(xd, pw, groups=2)
    assert y.shape == (n, t, c)
    h = n // 2
    assert rel(y[:h].float(), xf[:h] @ w0.t()) < TOL[dtype]
    assert rel(y[h:].float(), xf[h:] @ w1.t()) < TOL[dtype]
    # the same through a per-matrix launch (another kernel may run it: sums in another order, a few flipped roundings)
    y0 = ops.linear(xd[:h].contiguous(), ops.pack_linear(w0, None, dtype, "cuda"), split_k=1)
    assert rel(y[:h].float(), y0.float()) < 0.1 * TOL[dtype]


# (images, tokens, c, n_out, geglu, what)
LN_FOLD_CASES = [
    (16, 1024, 640, 1920, False, "256 x 320 ring tiles, register epilogue: norm1 -> [to_q; to_k; to_v] at the 32 x 32 level"),
    (4, 1024, 640, 1280, True, "the same tiles with the GEGLU register epilogue: norm3 -> GEGLU projection"),
    (4, 256, 1280, 3840, False, "128 x 160 tiles, 64-deep steps (one round of blocks): the 16 x 16 level"),
    (16, 256, 1280, 1280, False, "128 x 160 tiles, 32-deep steps, two blocks per CU"),
    (2, 100, 320, 960, False, "ragged tiles (200 rows): LDS-staged epilogue"),
    (2, 64, 320, 320, True, "a small problem: kept off the operands-in-registers kernel"),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", LN_FOLD_CASES, ids=[f"lnf{i}" for i in range(len(LN_FOLD_CASES))])
def test_igemm_layernorm_fold(ops, dtype, case):
    """mobi_igemm_params.ln_svec: Linear(LayerNorm(x)) as ONE launch on the raw rows (attention.py:234 `attn1(norm1(x))`, :264
    `ff(norm3(x))` of the reference) -- the packed matrix is W diag(gamma), the launch takes the row statistics from its own A
    fragments (v_dot2c in the MFMA shadow) and applies rstd (acc - mean s) + (W beta + b) in the accumulator domain.  Against
    fp32 torch's layer_norm + linear (+ GEGLU) on the same rounded inputs, and against the two-launch form of the engine."""
    n, t, c, n_out, geglu, _what = case
    xf, xd = rnd("lnf.x", (n, t, c), dtype, scale=1.5)
    xf = xf + 0.25                                          # a mean that is not zero
    xd = xf.to(dtype).cuda()
    xf = xd.float().cpu()
    rows = n_out * (2 if geglu else 1)
    w = torch.from_numpy(W.synth_param("lnf.w", (rows, c)))
    b = torch.from_numpy(W.synth_param("lnf.b", (rows,)))
    gamma = 1.0 + 0.3 * torch.from_numpy(W.synth_param("lnf.g", (c,))) * c ** 0.5 * 0.1
    beta = torch.from_numpy(W.synth_param("lnf.be", (c,))) * 2.0
    eps = 1e-5
    wf, bf = ops.fold_layernorm(w, b, gamma, beta)
    pack = ops.pack_geglu if geglu else ops.pack_linear
    pw = ops.with_row_sums(pack(wf, bf, dtype, "cuda"), eps)
    y = ops.linear(xd, pw)
    # reference: the folded layer as the engine rounds it (W diag(gamma) to the storage type), statistics in fp64
    wr = wf.to(dtype).double()
    x64 = xf.double()
    mean, var = x64.mean(-1, keepdim=True), x64.var(-1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    pre = rstd * (x64 @ wr.t() - mean * wr.sum(1)) + bf.double()
    ref = pre[..., :n_out] * F.gelu(pre[..., n_out:]) if geglu else pre
    assert y.shape == ref.shape
    assert rel(y.float(), ref.float()) < TOL[dtype]
    # the same layer the reference's way (LayerNorm, then the un-folded Linear): equal up to the storage type's rounding of the
    # normalised rows, which the fold does not have
    ln = F.layer_norm(x64, (c,), gamma.double(), beta.double(), eps)
    pre2 = ln @ w.double().t() + b.double()
    ref2 = pre2[..., :n_out] * F.gelu(pre2[..., n_out:]) if geglu else pre2
    assert rel(y.float(), ref2.float()) < 2 * TOL[dtype]
    y2 = ops.linear(ops.layernorm(xd, gamma.cuda(), beta.cuda(), eps), pack(w, b, dtype, "cuda"))
    assert rel(y2.float(), ref2.float()) < 2 * TOL[dtype]
