"""GPU parity of mobi_row_chain (csrc/chain.hip): programs of row-resident C -> C products against the same arithmetic in
fp32 torch on inputs pre-rounded to the storage type -- the launches between the attention kernels of a transformer block
(ldm/modules/attention.py:230-266 of the reference) -- and against the separate engine launches they replace."""
import pytest
import torch
import torch.nn.functional as F

from oracle import weights as W
from tests.test_gpu_ops import DT, TOL, rel, rnd

pytestmark = pytest.mark.gpu
C = 320


@pytest.fixture(scope="module")
def ops():
    from mobi_amd import ops as o
    return o


def _w(name, dtype, scale=1.0):
    w = torch.from_numpy(W.synth_param(name + ".weight", (C, C))) * scale
    b = torch.from_numpy(W.synth_param(name + ".bias", (C,)))
    return w, b


def _rt(t, dtype):
    return t.to(dtype).float()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,t", [(2, 128), (3, 384), (16, 4096)])
def test_chain_products_residual_store(ops, dtype, n, t):
    """load, product (+ per-image bias + residual -> row state, stored), two more products of the new row state."""
    name = f"chain.p.{n}.{t}"
    af, ad = rnd(name + ".a", (n, t, C), dtype)
    xf, xd = rnd(name + ".x", (n, t, C), dtype)
    w0, _ = _w(name + ".w0", dtype)
    wk, _ = _w(name + ".wk", dtype)
    wv, bv = _w(name + ".wv", dtype)
    rv = W.synth_input(name + ".rv", (n, C))
    cw0, cwk, cwv = (ops.pack_chain_weight(w, b, dtype, "cuda") for w, b in ((w0, None), (wk, None), (wv, bv)))
    x1 = torch.empty_like(xd)
    kv = torch.empty((n, t, 2 * C), device="cuda", dtype=dtype)
    prog = ops.ChainProgram().load(ad, "s").load(xd, "r")
    prog.product(cw0, resid=True, to_s=True, dst=x1, bias=rv.cuda().contiguous(), bias_img_stride=C)
    prog.product(cwk, dst=kv[..., :C]).product(cwv, dst=kv[..., C:])
    ops.row_chain([prog], n, t, dtype)
    torch.cuda.synchronize()
    r1 = F.linear(af, _rt(w0, dtype)) + rv[:, None, :] + xf
    assert rel(x1.float(), r1) < TOL[dtype]
    r1r = x1.float().cpu()                                              # the stored (rounded) rows feed the next products
    assert rel(kv[..., :C].float(), F.linear(r1r, _rt(wk, dtype))) < TOL[dtype]
    assert rel(kv[..., C:].float(), F.linear(r1r, _rt(wv, dtype), bv)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,t", [(2, 256), (4, 1024)])
def test_chain_layernorm_fold(ops, dtype, n, t):
    """LayerNorm folded into the projection: rs * (x (W diag gamma)^T) + cs * s + W beta against linear(layer_norm(x)) in
    fp32 torch, and against the engine's own layernorm -> linear pair (one rounding more)."""
    name = f"chain.f.{n}.{t}"
    xf, xd = rnd(name + ".x", (n, t, C), dtype, scale=2.0)
    xf = xf + 0.7                                                       # a mean the fold has to cancel
    xd = xf.to(dtype).cuda()
    xf = xd.float().cpu()
    w, b = _w(name + ".w", dtype)
    g = torch.from_numpy(W.synth_param(name + ".ln.weight", (C,)))
    bt = torch.from_numpy(W.synth_param(name + ".ln.bias", (C,)))
    cw = ops.pack_chain_weight(w, b, dtype, "cuda", ln=(g, bt), scale=0.25)
    q = torch.empty_like(xd)
    prog = ops.ChainProgram().load(xd, "s").rowstats(1e-5).product(cw, fold=True, dst=q)
    ops.row_chain([prog], n, t, dtype)
    ref = F.linear(F.layer_norm(xf, (C,), g, bt, 1e-5), w * 0.25, b * 0.25)
    assert rel(q.float(), ref) < TOL[dtype] * 1.5
    sep = ops.linear(ops.layernorm(xd, g.cuda(), bt.cuda(), 1e-5), ops.pack_linear(w * 0.25, b * 0.25, dtype, "cuda"))
    assert rel(q.float(), sep.float()) < TOL[dtype] * 2.5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,t,heads", [(2, 128, 8), (4, 512, 8), (2, 256, 5)])
def test_chain_adapter(ops, dtype, n, t, heads):
    """The two-key adapter as a chain operation against the formula in fp32 torch and against mobi_two_key_adapter."""
    name = f"chain.a.{n}.{t}.{heads}"
    xf, xd = rnd(name + ".x", (n, t, C), dtype, scale=2.0)
    a = W.synth_input(name + ".a", (n, heads, C)) * 0.05
    u = W.synth_input(name + ".u", (n, heads, C))
    b = W.synth_input(name + ".b", (n, C))
    cc = W.synth_input(name + ".c", (n, heads))
    mean = xf.mean(-1, keepdim=True)
    rstd = (xf.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    z = rstd * (torch.einsum("ntc,nhc->nth", xf, a) - mean * a.sum(-1)[:, None, :]) + cc[:, None, :]
    ref = xf + b[:, None, :] + torch.einsum("nth,nhc->ntc", torch.sigmoid(z), u)
    out = torch.empty_like(xd)
    tabs = (a.cuda(), a.sum(-1).contiguous().cuda(), cc.cuda(), u.cuda(), b.cuda(), 1e-5)
    image = ops.chain_adapter_image(tabs[0], tabs[2], tabs[3], tabs[4], dtype)
    ops.row_chain([ops.ChainProgram().load(xd, "s").adapter(dst=out)], n, t, dtype, adapter=(image, 1e-5))
    assert rel(out.float(), ref) < TOL[dtype]
    y = ops.two_key_adapter(xd, *tabs)
    assert rel(out.float(), y.float()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_chain_affine_and_two_programs(ops, dtype):
    """Per-image scale / shift on the loaded rows (GroupNorm folded to two vectors), then different programs on even and
    odd images writing to half-batch tensors (image index / 2) -- the camera / lidar split of the cross-modal step."""
    n, t = 4, 256
    name = "chain.2p"
    xf, xd = rnd(name + ".x", (n, t, C), dtype)
    sc, sh = W.synth_input(name + ".sc", (n, C)) * 0.3 + 1.0, W.synth_input(name + ".sh", (n, C)) * 0.2
    w0, b0 = _w(name + ".w0", dtype)
    w1, b1 = _w(name + ".w1", dtype)
    c0, c1 = ops.pack_chain_weight(w0, b0, dtype, "cuda"), ops.pack_chain_weight(w1, b1, dtype, "cuda")
    o0 = torch.empty((n // 2, t, C), device="cuda", dtype=dtype)
    o1 = torch.empty((n // 2, t, 2 * C), device="cuda", dtype=dtype)
    scd, shd = sc.cuda().contiguous(), sh.cuda().contiguous()
    p0 = ops.ChainProgram().load(xd, "s").affine(scd, shd).product(c0, dst=o0, dst_img_div=2)
    p1 = ops.ChainProgram().load(xd, "s").affine(scd, shd).product(c1, dst=o1[..., :C], dst_img_div=2).product(c0, dst=o1[..., C:], dst_img_div=2)
    ops.row_chain([p0, p1], n, t, dtype)
    xa = _rt(xf * sc[:, None, :] + sh[:, None, :], dtype)
    assert rel(o0.float(), F.linear(xa[0::2], _rt(w0, dtype), b0)) < TOL[dtype]
    assert rel(o1[..., :C].float(), F.linear(xa[1::2], _rt(w1, dtype), b1)) < TOL[dtype]
    assert rel(o1[..., C:].float(), F.linear(xa[1::2], _rt(w0, dtype), b0)) < TOL[dtype]


def test_chain_rejects_what_it_cannot_run(ops):
    from mobi_amd import _lib
    assert not ops.row_chain_supported(640, 1024) and not ops.row_chain_supported(320, 100) and ops.row_chain_supported(320, 1024)
    x = torch.zeros((2, 100, C), device="cuda", dtype=torch.float16)
    with pytest.raises(_lib.EngineError):
        ops.row_chain([ops.ChainProgram().load(x, "s").store(x)], 2, 100, torch.float16)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("n,side", [(6, 64), (32, 32)])
def test_transformer_block_chained_equals_unchained(dtype, n, side, monkeypatch):
    """BasicTransformerBlock at C = 320 (8 heads of 40) with the launches between its attention kernels chained
    (`_forward_chained`) against the one-by-one sequence on the same block, and the caller's tensor left untouched."""
    import mobi_amd
    from mobi_amd.ldm.modules import attention as A
    mobi_amd.set_engine_dtype(dtype)
    blk = A.BasicTransformerBlock(C, 8, 40, context_dim=768, bbox_cond=True, multimodal=True)
    W.fill_module_(blk, seed=41)
    blk = blk.cuda()
    t = side * side
    _, x = rnd(f"chain.blk.{n}.{side}", (n, t, C), dtype)
    ctx = W.synth_input(f"chain.blk.ctx.{n}", (n, 2, 768)).cuda()
    x0 = x.clone()
    monkeypatch.setattr(A, "ROW_CHAIN", True)
    monkeypatch.setattr(A, "ROW_CHAIN_MIN_ROWS", 1)
    assert blk._chain_ok(x, (None,), object())
    y1 = blk(x, context=ctx)
    assert torch.equal(x, x0)
    monkeypatch.setattr(A, "ROW_CHAIN", False)
    y0 = blk(x, context=ctx)
    assert bool(torch.isfinite(y1).all())
    assert rel(y1.float(), y0.float()) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_routing_thresholds_at_their_boundary_rows(dtype, monkeypatch):
    """The row-count switches of the transformer block (`FUSED_FF_MIN_ROWS`, `ROW_CHAIN_MIN_ROWS`, the two-key adapter's
    register kernel) at the boundary: 24,576 token rows (6 images of 64 x 64 -- the first count that takes the one-launch
    feed-forward and the row chains) and one 128-row tile below it must give the same function on either side of the switch."""
    import mobi_amd
    from mobi_amd import ops as O
    from mobi_amd.ldm.modules import attention as A
    mobi_amd.set_engine_dtype(dtype)
    assert A.FUSED_FF_MIN_ROWS == 24576 and A.ROW_CHAIN_MIN_ROWS == 24576
    blk = A.BasicTransformerBlock(C, 8, 40, context_dim=768, bbox_cond=True, multimodal=True)
    W.fill_module_(blk, seed=43)
    blk = blk.cuda()
    ff = blk.ff
    for rows in (24576, 24576 - 128):
        _, x = rnd(f"thr.ff.{rows}", (1, rows, C), dtype)
        fused = rows >= A.FUSED_FF_MIN_ROWS
        y = ff(x, residual=x, norm=blk.norm3)
        monkeypatch.setattr(A, "FUSED_FF_MIN_ROWS", 1 if not fused else 1 << 30)          # the other side of the switch
        y2 = ff(x, residual=x, norm=blk.norm3)
        monkeypatch.setattr(A, "FUSED_FF_MIN_ROWS", 24576)
        assert rel(y.float(), y2.float()) < 2 * TOL[dtype], rows
    for n, t in ((6, 4096), (2, 4096)):                                                  # 24,576 rows: chained; 8,192: one by one
        _, x = rnd(f"thr.blk.{n}", (n, t, C), dtype)
        ctx = W.synth_input(f"thr.ctx.{n}", (n, 2, 768)).cuda()
        assert O.two_key_adapter_fuses_ln(C, n * t)
        y = blk(x, context=ctx)
        monkeypatch.setattr(A, "ROW_CHAIN_MIN_ROWS", 1 if n * t < 24576 else 1 << 30)
        y2 = blk(x, context=ctx)
        monkeypatch.setattr(A, "ROW_CHAIN_MIN_ROWS", 24576)
        assert rel(y.float(), y2.float()) < 2 * TOL[dtype], n


@pytest.mark.parametrize("dtype", DT)
def test_spatial_transformer_pre_chain_equals_launches(dtype, monkeypatch):
    """SpatialTransformer at C = 320 with GroupNorm -> proj_in -> norm1 -> q | k | v as one chain launch (+ the statistics pass)
    against the four launches, same module, same input; the GroupNorm scale / shift vectors against torch."""
    import mobi_amd
    from mobi_amd import ops as O
    from mobi_amd.ldm.modules import attention as A
    mobi_amd.set_engine_dtype(dtype)
    st = A.SpatialTransformer(C, 8, 40, depth=1, context_dim=768, bbox_cond=True, multimodal=True)
    W.fill_module_(st, seed=47)
    st = st.cuda()
    n, side = 6, 64
    xf, x = rnd("chain.st.x", (n, side, side, C), dtype, 1.5)
    ctx = W.synth_input("chain.st.ctx", (n, 2, 768)).cuda()
    g, b = st.norm.affine()
    sc, sh = O.groupnorm_scale_shift(x, g, b, st.norm.eps)
    ref = F.group_norm(xf.permute(0, 3, 1, 2), 32, g.cpu(), b.cpu(), st.norm.eps).permute(0, 2, 3, 1)
    got = xf * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :]
    assert rel(got, ref) < 1e-5
    monkeypatch.setattr(A, "PRE_CHAIN", True)
    assert st._pre_chain_ok(x)
    y1 = st(x, context=ctx)
    monkeypatch.setattr(A, "PRE_CHAIN", False)
    y0 = st(x, context=ctx)
    assert bool(torch.isfinite(y1).all()) and rel(y1.float(), y0.float()) < 2 * TOL[dtype]
