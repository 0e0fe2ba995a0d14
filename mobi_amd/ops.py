"""Thin host wrappers: torch tensors (device memory + stream plumbing) -> C ABI calls.

No arithmetic happens here.  Activations are channels-last tensors
`[N, H, W, C]` (tokens `[N, T, C]` are the same memory) in float16 / bfloat16;
a batch-strided view such as `x[::2]` is passed in place through the image
stride.  Every function raises if the HIP library is unavailable or the tensor
is not on a ROCm device -- there is no CPU path.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_SILU, EPI_GEGLU, EPI_NONE, MOBI_BF16, MOBI_F16, OUT_ROWS, OUT_ROWS_F32,
                   OUT_TRANSPOSED)


# optional launch profiler (bench.py): a list receiving (kind, algorithmic_flops, start_event, end_event)
_PROFILE = None


def set_profiler(sink):
    global _PROFILE
    _PROFILE = sink


class _Timed:
    """Brackets one launch with events on the launch stream when a profiler is installed."""

    def __init__(self, kind, flops, nbytes=0.0, tag=""):
        self.kind, self.flops, self.nbytes, self.tag = kind, flops, nbytes, tag

    def __enter__(self):
        if _PROFILE is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _PROFILE is not None:
            self.e1.record()
            _PROFILE.append((self.kind, self.flops, self.e0, self.e1, self.nbytes, self.tag))
        return False


# --------------------------------------------------------------------------------------
# stream-level concurrency: independent launches of one block (e.g. the q / k / v projections) run on
# side HIP streams so that their latency-bound prologues / epilogues overlap; fork waits on the caller's
# stream, join makes the caller's stream wait on every side stream.  Graph-capture safe (events only).
# --------------------------------------------------------------------------------------
_SIDE = {}
_CONCURRENT = os.environ.get("MOBI_CONCURRENCY", "0") == "1"     # measured: SLOWER inside the step graph (fork / join edges: +0.2 ms per mobi_nusc_512 step, +0.35 per mobi_nusc_256 step, profiles/r05_ab_graph_branches.txt)


def set_concurrency(flag):
    global _CONCURRENT
    _CONCURRENT = bool(flag)


def concurrently(*fns):
    """Run independent launch sequences concurrently; returns their results in order."""
    if not _CONCURRENT or _PROFILE is not None or len(fns) == 1:
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    key = (main.device, main.cuda_stream)
    pool = _SIDE.setdefault(key, [])
    while len(pool) < len(fns) - 1:
        pool.append(torch.cuda.Stream(device=main.device))
    start = torch.cuda.Event()
    start.record(main)
    results = [None] * len(fns)
    for i, f in enumerate(fns[1:]):
        side = pool[i]
        side.wait_event(start)
        with torch.cuda.stream(side):
            results[i + 1] = f()
    results[0] = fns[0]()
    for i in range(len(fns) - 1):
        main.wait_stream(pool[i])
    return results


def _dt(t):
    if t == torch.float16:
        return MOBI_F16
    if t == torch.bfloat16:
        return MOBI_BF16
    raise TypeError(f"engine storage type must be float16 or bfloat16, got {t}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t):
    if not t.is_cuda:
        raise _lib.EngineUnavailable("MObI engine tensors must live on a ROCm device (no CPU path)")
    return t


def _ptr(t):
    return None if t is None else C.c_void_p(_dev(t).data_ptr())


def _img_stride(t):
    """elements between images; inner dims must be dense."""
    inner = t[0]
    if not inner.is_contiguous():
        raise ValueError("activation tensor must be dense within an image")
    return t.stride(0) if t.shape[0] > 1 else inner.numel()


# --------------------------------------------------------------------------------------
# weight packing (load time)
# --------------------------------------------------------------------------------------
@dataclass
class Packed:
    w: torch.Tensor                 # T [n_packed, kh*kw*cin]
    bias: Optional[torch.Tensor]    # f32
    kh: int
    kw: int
    cin: int
    cout: int
    n_packed: int
    geglu: bool = False
    k_order: int = 0                # 0: k = tap*C + c; 1: 64-channel-chunk major (mobi_igemm_params.k_order)
    wt: Optional[torch.Tensor] = None   # the same matrix as 1-KiB request images (mobi_igemm_params.weight_tiled)
    svec: Optional[torch.Tensor] = None  # LayerNorm folded in (fold_layernorm): f32 [n_packed] row sums of the rounded W diag(gamma)
    ln_eps: float = 0.0

    def __post_init__(self):
        if self.wt is None and self.k_order == 0 and self.w.dim() == 2 and self.w.is_cuda and TILED_WEIGHTS:
            self.wt = tile_weights(self.w)


TILED_WEIGHTS = True
TILE_WEIGHTS_TORCH = False                # tests: the torch restatement of mobi_tile_weights
_RING_PERM = (0, 2, 3, 1)              # 16-byte slot permutation of the ring kernels' LDS rows, per (row >> 2) & 3


def tile_weights(w):
    """T [n, k] (n % 16 == 0, k % 32 == 0) -> the same values as [n / 16][k / 32] blocks of 1 KiB, each block the LDS
    image of a 16-row x 32-deep piece (row r holds its four 16-byte chunks c at slot c ^ P[(r >> 2) & 3]): ONE contiguous
    KiB per LDS-DMA request instead of sixteen 64-byte row segments.  The request path of a CU takes 57-63 B per clock of
    the former and 21-28 of the latter (tools/probes/lds_dma_rate.hip)."""
    n, k = w.shape
    if n % 16 or k % 32:
        return None
    if w.is_cuda and w.is_contiguous() and w.element_size() == 2 and not TILE_WEIGHTS_TORCH:
        out = torch.empty((n // 16, k // 32, 16, 4, 8), device=w.device, dtype=w.dtype)
        _lib.check(_lib.load().mobi_tile_weights(_ptr(w), _ptr(out), n, k, _stream()), "mobi_tile_weights")
        return out
    v = w.reshape(n // 16, 16, k // 32, 4, 8)                                   # [piece, row, step, chunk, 8]
    r = torch.arange(16, device=w.device)
    perm = torch.tensor(_RING_PERM, device=w.device)[(r >> 2) & 3]              # [16]
    src = torch.arange(4, device=w.device)[None, :] ^ perm[:, None]             # slot s of row r holds chunk s ^ P
    v = v.permute(0, 2, 1, 3, 4)                                                # [piece, step, row, chunk, 8]
    idx = src[None, None, :, :, None].expand(v.shape[0], v.shape[1], 16, 4, 8)
    return torch.gather(v, 3, idx).contiguous()


def pack_conv(weight, bias, dtype, device, chunk_major=False):
    """OIHW fp32 -> [O][kh*kw*I], the layout mobi_igemm / conv_small_cout read.
    Default k = tap*C + c.  chunk_major (igemm only, I % 64 == 0, more than one tap): k runs over
    (64-channel chunk, tap, channel in chunk) so that consecutive k-tiles revisit the same pixels (L2 reuse)."""
    o, i, kh, kw = weight.shape
    w = weight.detach().to(device=device, dtype=torch.float32)
    b = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
    if chunk_major and i % 64 == 0 and kh * kw > 1:
        w = w.reshape(o, i // 64, 64, kh, kw).permute(0, 1, 3, 4, 2).reshape(o, kh * kw * i)
        return Packed(w.to(dtype).contiguous(), b, kh, kw, i, o, o, k_order=1)
    w = w.permute(0, 2, 3, 1).reshape(o, kh * kw * i)
    return Packed(w.to(dtype).contiguous(), b, kh, kw, i, o, o)


def pack_conv_padded_cin(weight, bias, dtype, device, cin_pad=32):
    """Thin-input convs (9, 4, 3, 2 input channels): zero-pad the input-channel axis to `cin_pad` so the
    matrix-core kernel can run them on sources packed by `pack_sources`."""
    o, i, kh, kw = weight.shape
    w = torch.zeros((o, cin_pad, kh, kw), dtype=torch.float32, device=device)
    w[:, :i] = weight.detach().to(device=device, dtype=torch.float32)
    return pack_conv(w, bias, dtype, device)


def pack_linear(weight, bias, dtype, device):
    return pack_conv(weight[:, :, None, None], bias, dtype, device)


def fold_layernorm(weight, bias, gamma, beta):
    """Linear(LayerNorm(x)) with the LayerNorm's affine part folded into the layer (exact algebra, fp64 on the host, load-time
    work like the other packs): LN(x) W^T + b = rstd (x (W diag gamma)^T - mean s) + (W beta + b).
    weight [n, k], bias [n] | None, gamma / beta [k] -> (W diag(gamma) fp32, W beta + b fp32); the row sums `s` are taken from the
    ROUNDED packed matrix (`with_row_sums`), mobi_igemm applies rstd / mean per row (mobi_igemm_params.ln_svec)."""
    wd = weight.detach().double().cpu()
    g, b = gamma.detach().double().cpu(), beta.detach().double().cpu()
    bd = wd @ b + (0.0 if bias is None else bias.detach().double().cpu())
    return (wd * g[None, :]).float(), bd.float()


def with_row_sums(pw: Packed, eps):
    """The pack of a LayerNorm-folded layer: + the fp32 row sums of the matrix as the kernels will read it (rounded, packed order)."""
    pw.svec = pw.w.double().sum(dim=1).float().contiguous()
    pw.ln_eps = float(eps)
    return pw


def geglu_layout(inner):
    """(unit, n_packed) of the packed GEGLU projection: every 16-row MFMA tile of the packed matrix holds
    `unit` = 8 value rows followed by the 8 gate rows of the same outputs, so that a value and its gate sit in
    lanes 32 apart of one accumulator tile (one v_permlane32_swap pairs them in the register epilogue)."""
    if inner % 8:
        raise ValueError("GEGLU inner width must be a multiple of 8")
    return 8, 2 * inner


def pack_geglu(weight, bias, dtype, device):
    """GEGLU.proj (attention.py:38-46): rows [0, inner) are the value half, [inner, 2*inner) the gate."""
    two_inner, cin = weight.shape
    inner = two_inner // 2
    unit, n_packed = geglu_layout(inner)
    w = weight.detach().to(device=device, dtype=torch.float32)
    b = bias.detach().to(device=device, dtype=torch.float32)
    t = inner // unit
    wp = torch.stack([w[:inner].reshape(t, unit, cin), w[inner:].reshape(t, unit, cin)], dim=1).reshape(n_packed, cin)
    bp = torch.stack([b[:inner].reshape(t, unit), b[inner:].reshape(t, unit)], dim=1).reshape(n_packed)
    return Packed(wp.to(dtype).contiguous(), bp.contiguous(), 1, 1, cin, inner, n_packed, geglu=True)


class PackedFf:
    """Chunk images of a GEGLU feed-forward pair for mobi_ff_geglu (layout: include/mobi_engine.h)."""

    def __init__(self, buf, b2, c, hidden, dtype):
        self.buf, self.b2, self.c, self.hidden, self.dtype = buf, b2, c, hidden, dtype


def ff_geglu_supported(c, hidden):
    return c == 320 and hidden % 32 == 0


def pack_ff_geglu(w1, b1, w2, b2, dtype, device):
    """w1: GEGLU.proj.weight [2 hidden, c] (value rows, then gate rows), b1 [2 hidden]; w2: net[2].weight [c, hidden],
    b2 [c] -> PackedFf.  Index arithmetic only (a gather of the fp32 masters, one rounding to the storage type)."""
    two_h, c = w1.shape
    hidden = two_h // 2
    assert w2.shape == (c, hidden) and ff_geglu_supported(c, hidden)
    ks_n, mt_n, nch = c // 16, c // 32, hidden // 32
    w1 = w1.detach().to(device=device, dtype=torch.float32)
    w2 = w2.detach().to(device=device, dtype=torch.float32)
    ar = lambda n: torch.arange(n, device=device)
    lane, j = ar(64), ar(8)
    m_, h_ = lane & 31, lane >> 5
    # first product: fragment (t, ks): rows t hidden + 32 chunk + m, columns 16 ks + 8 h + j
    rows1 = (ar(2)[None, :, None, None] * hidden + ar(nch)[:, None, None, None] * 32 + m_[None, None, None, :])   # [nch,2,1,64]
    cols1 = (ar(ks_n)[:, None, None] * 16 + h_[None, :, None] * 8 + j[None, None, :])                             # [KS,64,8]
    img1 = w1[rows1.expand(nch, 2, ks_n, 64)[..., None], cols1[None, None]]                                      # [nch,2,KS,64,8]
    # second product: fragment (m, s): rows 32 m + m_, units 32 chunk + 16 s + 8 (j >> 2) + 4 h + (j & 3)
    rows2 = (ar(mt_n)[:, None, None] * 32 + m_[None, None, :])                                                    # [MT,1,64]
    unit = (ar(2)[:, None, None] * 16 + ((j >> 2) * 8 + (j & 3))[None, None, :] + h_[None, :, None] * 4)           # [2,64,8]
    cols2 = ar(nch)[:, None, None, None, None] * 32 + unit[None, None]                                            # [nch,1,2,64,8]
    img2 = w2[rows2.expand(mt_n, 2, 64)[None, ..., None], cols2.expand(nch, mt_n, 2, 64, 8)]                      # [nch,MT,2,64,8]
    esz = torch.empty((), dtype=dtype).element_size()
    assert esz == 2
    p1 = img1.to(dtype).contiguous().view(torch.uint8).reshape(nch, -1)
    p2 = img2.to(dtype).contiguous().view(torch.uint8).reshape(nch, -1)
    b1f = b1.detach().to(device=device, dtype=torch.float32)
    bias = torch.zeros((nch, 256), device=device, dtype=torch.float32)
    bias[:, :32] = b1f[:hidden].reshape(nch, 32)
    bias[:, 32:64] = b1f[hidden:].reshape(nch, 32)
    buf = torch.cat([p1.reshape(-1), torch.cat([p2, bias.view(torch.uint8).reshape(nch, -1)], dim=1).reshape(-1)]).contiguous()
    assert buf.numel() == _lib.load().mobi_ff_geglu_packed_bytes(c, hidden)
    return PackedFf(buf, None if b2 is None else b2.detach().to(device=device, dtype=torch.float32).contiguous(), c, hidden, dtype)


def ff_geglu(x, pf: PackedFf, residual=None, out=None, ln=None):
    """x: T [..., c] dense -> out T [..., c] = GEGLU feed-forward (+ residual) in one launch (mobi_ff_geglu).
    ln = (gamma, beta, eps): the feed-forward of LayerNorm(x) (the kernel normalises the rows it holds in registers)."""
    lib = _lib.load()
    _dev(x)
    assert x.is_contiguous() and x.shape[-1] == pf.c and x.dtype == pf.dtype
    rows = x.numel() // pf.c
    if out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.shape == x.shape and (residual is None or (residual.is_contiguous() and residual.shape == x.shape))
    p = _lib.FfGegluParams()
    p.x, p.rows, p.c, p.hidden, p.w_packed, p.b2 = _ptr(x), rows, pf.c, pf.hidden, _ptr(pf.buf), _ptr(pf.b2)
    p.residual, p.out, p.dtype = _ptr(residual), _ptr(out), _dt(x.dtype)
    if ln is not None:
        g, b, eps = ln
        assert g.dtype == b.dtype == torch.float32 and g.numel() == b.numel() == pf.c and g.is_contiguous() and b.is_contiguous()
        p.ln_gamma, p.ln_beta, p.ln_eps = _ptr(g), _ptr(b), eps
    fl = 2.0 * rows * pf.c * (2 * pf.hidden) + 2.0 * rows * pf.hidden * pf.c
    reads = 1 + (residual is not None and (ln is None or residual.data_ptr() != x.data_ptr()))
    with _Timed("ff_geglu", fl, 2.0 * rows * pf.c * (reads + 1), f"rows={rows} c={pf.c} hidden={pf.hidden}"):
        _lib.check(lib.mobi_ff_geglu(C.byref(p), _stream()), "mobi_ff_geglu")
    return out


# --------------------------------------------------------------------------------------
# row-resident chains (mobi_row_chain, csrc/chain.hip)
# --------------------------------------------------------------------------------------
class ChainWeight:
    """One product of a chain: the [320][320] matrix as a 200-KiB chunk image (layout: include/mobi_engine.h), its fp32 bias
    (zeros where the layer has none) and, for a LayerNorm-folded projection, the row sums of the rounded W diag(gamma)."""

    def __init__(self, image, bias, svec=None):
        self.image, self.bias, self.svec = image, bias, svec


def row_chain_supported(channels, rows_per_image):
    return bool(_lib.load().mobi_row_chain_supported(int(channels), int(rows_per_image)))


def pack_chain_weight(w, bias, dtype, device, ln=None, scale=1.0):
    """w: [320, 320] (out, in) fp32 master; bias [320] or None; ln = (gamma, beta): the LayerNorm in front of the layer folded
    in (W' = scale W diag(gamma), svec = row sums of the ROUNDED W', bias' = scale (W beta) + bias -- fp64 on the host,
    load-time work like the other packs); scale multiplies the layer (to_q's scale * log2 e) -> ChainWeight."""
    n, k = w.shape
    assert n == k == 320
    wd = w.detach().double().cpu() * scale
    bd = torch.zeros(n, dtype=torch.float64) if bias is None else bias.detach().double().cpu() * scale
    svec = None
    if ln is not None:
        gamma, beta = (t.detach().double().cpu() for t in ln)
        bd = bd + wd @ beta
        wd = wd * gamma[None, :]
    wr = wd.float().to(dtype)                                   # the one rounding to the storage type
    if ln is not None:
        svec = wr.double().sum(dim=1).float().contiguous().to(device)
    ar = torch.arange
    lane, j = ar(64), ar(8)
    i = lane & 31
    tau = (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1)
    # [chunk c][kk][m][lane][j]: row 32 m + tau(lane & 31), column 16 (2 c + kk) + 8 (lane >> 5) + j
    rows = (ar(10)[None, None, :, None, None] * 32 + tau[None, None, None, :, None]).expand(10, 2, 10, 64, 8)
    cols = ((2 * ar(10)[:, None, None, None, None] + ar(2)[None, :, None, None, None]) * 16
            + (lane >> 5)[None, None, None, :, None] * 8 + j[None, None, None, None, :]).expand(10, 2, 10, 64, 8)
    img = wr[rows, cols].contiguous().view(torch.uint8).reshape(-1).to(device)
    assert img.numel() == _lib.load().mobi_row_chain_weight_bytes(320)
    return ChainWeight(img, bd.float().contiguous().to(device), svec)


def groupnorm_scale_shift(x, gamma, beta, eps):
    """x: T [N, T, C] or [N, H, W, C] dense -> (scale, shift) fp32 [N, C]: GroupNorm(32 groups) as y = x * scale + shift."""
    lib = _lib.load()
    assert x.is_contiguous()
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    scale = torch.empty((n, c), device=x.device, dtype=torch.float32)
    shift = torch.empty((n, c), device=x.device, dtype=torch.float32)
    _lib.check(lib.mobi_groupnorm_scale_shift(_ptr(x), _ptr(gamma), _ptr(beta), eps, _ptr(scale), _ptr(shift), n, hw, c, _dt(x.dtype),
                                              _stream()), "mobi_groupnorm_scale_shift")
    return scale, shift


class ChainProgram:
    """A list of mobi_chain_op; the methods mirror the operation codes (include/mobi_engine.h): the row state `s` (every
    product's operand; what adapter / rowstats / store work on) and a residual `r` that a `resid` product consumes."""

    def __init__(self):
        self.ops, self.keep = [], []

    def _op(self, code, flags=0):
        o = _lib.ChainOp()
        o.code, o.flags = code, flags
        self.ops.append(o)
        return o

    @staticmethod
    def _rows(t):
        """[N, T, >= C] view, channel stride 1 -> (ptr, image stride, row stride)."""
        assert t.dim() == 3 and t.stride(2) == 1
        return _ptr(t), t.stride(0), t.stride(1)

    def load(self, t, which="s", img_div=1):
        o = self._op(_lib.CH_LOAD_S if which == "s" else _lib.CH_LOAD_R)
        o.p0, o.img_stride, o.row_stride = self._rows(t)
        o.img_div = img_div
        self.keep.append(t)
        return self

    def affine(self, scale, shift):
        o = self._op(_lib.CH_AFFINE_S)
        assert scale.dtype == shift.dtype == torch.float32 and scale.is_contiguous() and shift.is_contiguous()
        o.bias, o.svec = _ptr(scale), _ptr(shift)
        self.keep += [scale, shift]
        return self

    def rowstats(self, eps):
        self._op(_lib.CH_ROWSTATS).eps = eps
        return self

    def _dst(self, o, dst, dst_img_div):
        o.dst, o.dst_img_stride, o.dst_row_stride = self._rows(dst)
        o.dst_img_div = dst_img_div
        self.keep.append(dst)

    def product(self, cw: ChainWeight, *, fold=False, resid=False, to_s=False, dst=None, dst_img_div=1, bias=None,
                bias_img_stride=0, bias_img_div=1):
        """bias: overrides the packed one (a per-image fp32 [images, 320] vector with bias_img_stride = 320)."""
        fl = (_lib.CH_FOLD if fold else 0) | (_lib.CH_RESID if resid else 0) | (_lib.CH_TO_S if to_s else 0) | \
             (_lib.CH_STORE if dst is not None else 0)
        o = self._op(_lib.CH_PRODUCT, fl)
        o.p0 = _ptr(cw.image)
        b = cw.bias if bias is None else bias
        assert b.dtype == torch.float32 and b.is_contiguous()
        o.bias, o.bias_img_stride, o.bias_img_div = _ptr(b), bias_img_stride, bias_img_div
        if fold:
            assert cw.svec is not None
            o.svec = _ptr(cw.svec)
        if dst is not None:
            self._dst(o, dst, dst_img_div)
        self.keep += [cw, b]
        return self

    def adapter(self, dst, dst_img_div=1):
        self._dst(self._op(_lib.CH_ADAPTER, _lib.CH_STORE), dst, dst_img_div)
        return self

    def store(self, dst, dst_img_div=1):
        self._dst(self._op(_lib.CH_STORE_S), dst, dst_img_div)
        return self

    def finish(self):
        """Chain the prefetch pointers: every product names the next product's image."""
        prods = [o for o in self.ops if o.code == _lib.CH_PRODUCT]
        for o, nxt in zip(prods, prods[1:] + [None]):
            o.p1 = None if nxt is None else nxt.p0
        return self


def chain_adapter_image(a, c, u, b, dtype, out=None):
    """The two-key adapter's tables (a, u fp32 [N, H, 320]; c [N, H]; b [N, 320]: ops.two_key_adapter's) as the LDS images
    mobi_row_chain's ADAPTER operation copies in -> uint8 [N, bytes] (mobi_row_chain_adapter_image)."""
    lib = _lib.load()
    n, h, ch = a.shape
    for tns in (a, c, u, b):
        assert tns.dtype == torch.float32 and tns.is_contiguous()
    assert u.shape == (n, h, ch) and c.shape == (n, h) and b.shape == (n, ch)
    nb = lib.mobi_row_chain_adapter_image_bytes(ch)
    assert nb > 0
    if out is None:
        out = torch.empty((n, nb), device=a.device, dtype=torch.uint8)
    _lib.check(lib.mobi_row_chain_adapter_image(_ptr(a), _ptr(c), _ptr(u), _ptr(b), n, h, ch, _dt(dtype), _ptr(out), _stream()),
               "mobi_row_chain_adapter_image")
    return out


def row_chain(programs, images, rows_per_image, dtype, adapter=None, flops=0.0, nbytes=0.0, note=""):
    """Run one or two ChainPrograms (two: even images run the first, odd images the second) over `images` x
    `rows_per_image` token rows of 320 channels.  adapter = (image from chain_adapter_image, eps)."""
    lib = _lib.load()
    p = _lib.RowChainParams()
    p.dtype, p.channels, p.images, p.rows_per_image, p.nprog = _dt(dtype), 320, images, rows_per_image, len(programs)
    for k, prog in enumerate(programs):
        prog.finish()
        assert 0 < len(prog.ops) <= _lib.CHAIN_MAX_OPS
        p.nops[k] = len(prog.ops)
        for i, o in enumerate(prog.ops):
            p.prog[k][i] = o
    if adapter is not None:
        image, eps = adapter
        assert image.dtype == torch.uint8 and image.is_contiguous() and image.shape[0] == images
        p.ad_image, p.ad_eps = _ptr(image), eps
    with _Timed("row_chain", flops, nbytes, note):
        _lib.check(lib.mobi_row_chain(C.byref(p), _stream()), "mobi_row_chain")


# --------------------------------------------------------------------------------------
# backward pass (first slice of the training step: csrc/backward.hip)
# --------------------------------------------------------------------------------------
def trunk_add(trunk, inc, dtype):
    """trunk: fp32 dense tensor, updated in place: trunk += inc (T, same shape; None: no update) -> its 16-bit copy."""
    lib = _lib.load()
    assert trunk.dtype == torch.float32 and trunk.is_contiguous() and (inc is None or (inc.is_contiguous() and inc.shape == trunk.shape))
    x16 = torch.empty(trunk.shape, device=trunk.device, dtype=dtype)
    _lib.check(lib.mobi_trunk_add(_ptr(trunk), _ptr(inc), _ptr(x16), trunk.numel(), _dt(dtype), _stream()), "mobi_trunk_add")
    return x16


def split_f32(x32, dtype, parts=2):
    """fp32 [..., C] dense -> T [..., parts * C] = hi | lo (| hi): the operand form of groupnorm's GN_OUT_SPLIT / GN_OUT_SPLIT3 outputs."""
    lib = _lib.load()
    assert x32.dtype == torch.float32 and x32.is_contiguous() and parts in (2, 3)
    c = x32.shape[-1]
    out = torch.empty(tuple(x32.shape[:-1]) + (parts * c,), device=x32.device, dtype=dtype)
    _lib.check(lib.mobi_split_f32(_ptr(x32), _ptr(out), x32.numel() // c, c, parts, _dt(dtype), _stream()), "mobi_split_f32")
    return out


def transpose(x):
    """T [rows, cols] (row stride free) -> T [cols, rows] dense."""
    lib = _lib.load()
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    out = torch.empty((cols, rows), device=x.device, dtype=x.dtype)
    _lib.check(lib.mobi_transpose(_ptr(x), x.stride(0), _ptr(out), rows, cols, _dt(x.dtype), _stream()), "mobi_transpose")
    return out


def colsum(dy):
    """T [rows, cols] (row stride free) -> fp32 [cols] (a bias gradient; fixed summation order)."""
    lib = _lib.load()
    assert dy.dim() == 2 and dy.stride(1) == 1
    rows, cols = dy.shape
    part = torch.empty((lib.mobi_backward_partial_blocks(rows) + 64, cols), device=dy.device, dtype=torch.float32)
    out = torch.empty(cols, device=dy.device, dtype=torch.float32)
    _lib.check(lib.mobi_colsum(_ptr(dy), dy.stride(0), rows, cols, _dt(dy.dtype), _ptr(part), _ptr(out), _stream()), "mobi_colsum")
    return out


def linear_wgrad(dy, x):
    """dW = dy^T x for y = x W^T: dy T [rows, n], x T [rows, k] (row strides free; rows % 32 == 0) -> fp32 [n, k].
    Both operands are transposed (mobi_transpose) and multiplied on mobi_igemm with the token axis as k."""
    rows, n = dy.shape
    k = x.shape[1]
    assert x.shape[0] == rows
    if rows % 32:
        # a handful of token rows (the 1 x 1 ... 4 x 4 levels of a small latent): fp32 FMA chains instead of the matrix cores
        return linear_f32(dy.float().t().contiguous(), x.float().t().contiguous())
    dyt, xt = transpose(dy), transpose(x)                                    # [n, rows], [k, rows]
    pw = Packed(xt, None, 1, 1, rows, k, k)
    return igemm(dyt.view(1, 1, n, rows), pw, out_mode=OUT_ROWS_F32).view(n, k)


def layernorm_bwd(x, dy, gamma, eps, dx_add=None):
    """x, dy: T [N, T, C] (x may be a batch-strided view with dense rows) -> (dx T [N, T, C], d gamma fp32 [C], d beta fp32 [C]);
    dx_add: T [N, T, C] dense, added to dx (the gradient arriving over the residual branch)."""
    lib = _lib.load()
    n, t, c = x.shape
    x2, dy2 = x.reshape(n * t, c), dy.reshape(n * t, c)                       # (views when dense, copies when strided)
    dx = torch.empty((n, t, c), device=x.device, dtype=x.dtype)
    nb = lib.mobi_backward_partial_blocks(n * t) + 64
    part = torch.empty((nb, 2, c), device=x.device, dtype=torch.float32)
    dgb = torch.empty((2, c), device=x.device, dtype=torch.float32)
    p = _lib.LayerNormBwdParams()
    p.x, p.dy, p.x_row_stride, p.dy_row_stride = _ptr(x2), _ptr(dy2), x2.stride(0), dy2.stride(0)
    p.gamma, p.eps = _ptr(gamma), eps
    if dx_add is not None:
        assert dx_add.is_contiguous() and dx_add.shape == (n, t, c)
        p.dx_add = _ptr(dx_add)
    p.dx, p.partial, p.dgamma_dbeta, p.rows, p.channels, p.dtype = _ptr(dx), _ptr(part), _ptr(dgb), n * t, c, _dt(x.dtype)
    _lib.check(lib.mobi_layernorm_bwd(C.byref(p), _stream()), "mobi_layernorm_bwd")
    return dx, dgb[0], dgb[1]


def groupnorm_bwd(x, dy, gamma, beta, eps, silu, dx_add=None, one_block_per_group=False):
    """x, dy: T [N, H, W, C] dense -> dx (+ dx_add) of GroupNorm(32 groups)(+ SiLU); the affine parameters are frozen."""
    lib = _lib.load()
    n, h, w, c = x.shape
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape and (dx_add is None or dx_add.is_contiguous())
    dx = torch.empty_like(x)
    ws = None if one_block_per_group else torch.empty(lib.mobi_groupnorm_bwd_workspace_floats(n, h * w, c), device=x.device, dtype=torch.float32)
    _lib.check(lib.mobi_groupnorm_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), eps, int(silu), _ptr(dx_add), _ptr(dx), n, h * w, c,
                                      _dt(x.dtype), _ptr(ws), _stream()), "mobi_groupnorm_bwd")
    return dx


def sumpool2(src):
    """T [N, 2h, 2w, C] -> T [N, h, w, C]: sums of the 2 x 2 pixel blocks (backward of nearest x2 upsampling)."""
    lib = _lib.load()
    n, h2, w2, c = src.shape
    assert src.is_contiguous() and h2 % 2 == 0 and w2 % 2 == 0
    out = torch.empty((n, h2 // 2, w2 // 2, c), device=src.device, dtype=src.dtype)
    _lib.check(lib.mobi_sumpool2(_ptr(src), _ptr(out), n, h2 // 2, w2 // 2, c, _dt(src.dtype), _stream()), "mobi_sumpool2")
    return out


def add(a, b):
    """T + T -> T (dense, same shape)."""
    lib = _lib.load()
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype
    out = torch.empty_like(a)
    _lib.check(lib.mobi_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), _dt(a.dtype), _stream()), "mobi_add")
    return out


def silu_bwd_f32(z, dy):
    lib = _lib.load()
    assert z.dtype == dy.dtype == torch.float32 and z.is_contiguous() and dy.is_contiguous() and z.shape == dy.shape
    dx = torch.empty_like(z)
    _lib.check(lib.mobi_silu_bwd_f32(_ptr(z), _ptr(dy), _ptr(dx), z.numel(), _stream()), "mobi_silu_bwd_f32")
    return dx


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
    """One AdamW update of an fp32 parameter tensor IN PLACE (raw-pointer write: the caller bumps the tensor's version)."""
    lib = _lib.load()
    for t_ in (param, grad, exp_avg, exp_avg_sq):
        assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == param.numel()
    _lib.check(lib.mobi_adamw_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), lr, betas[0], betas[1], eps,
                                   weight_decay, step, _stream()), "mobi_adamw_step")


def geglu_fwd(pre):
    """pre: T [..., 2 inner] = [value | gate] dense -> value * gelu_erf(gate): T [..., inner]."""
    lib = _lib.load()
    assert pre.is_contiguous()
    inner = pre.shape[-1] // 2
    h = torch.empty(pre.shape[:-1] + (inner,), device=pre.device, dtype=pre.dtype)
    _lib.check(lib.mobi_geglu_fwd(_ptr(pre), _ptr(h), pre.numel() // (2 * inner), inner, _dt(pre.dtype), _stream()), "mobi_geglu_fwd")
    return h


def geglu_bwd(pre, dh):
    lib = _lib.load()
    assert pre.is_contiguous() and dh.is_contiguous()
    inner = pre.shape[-1] // 2
    dpre = torch.empty_like(pre)
    _lib.check(lib.mobi_geglu_bwd(_ptr(pre), _ptr(dh), _ptr(dpre), pre.numel() // (2 * inner), inner, _dt(pre.dtype), _stream()),
               "mobi_geglu_bwd")
    return dpre


def attention_bwd(q, k, v, o, dout, heads, scale, force_vector=False):
    """q: T [N, Tq, >= C] view, k / v: T [N, Tk, >= C] views (channel stride 1), o / dout: [N, Tq, C] -> (dq, dk, dv) dense T."""
    lib = _lib.load()
    n, tq, tk = q.shape[0], q.shape[1], k.shape[1]
    c = o.shape[2]
    new = lambda t_: torch.empty((n, t_, c), device=q.device, dtype=q.dtype)
    dq, dk, dv = new(tq), new(tk), new(tk)
    lse = torch.empty((2, n, heads, tq), device=q.device, dtype=torch.float32)
    p = _lib.AttentionBwdParams()
    for name, t_ in (("q", q), ("k", k), ("v", v), ("o", o), ("dout", dout)):
        assert t_.shape[2] == 1 or t_.stride(2) == 1
        setattr(p, name, _ptr(t_))
        setattr(p, name + "_img_stride", t_.stride(0))
        setattr(p, name + "_row_stride", t_.stride(1))
    p.dq, p.dk, p.dv, p.lse, p.dvec = _ptr(dq), _ptr(dk), _ptr(dv), _ptr(lse[0]), _ptr(lse[1])
    p.images, p.heads, p.dh, p.tq, p.tk, p.scale, p.dtype = n, heads, c // heads, tq, tk, scale, _dt(q.dtype)
    p.force_vector = int(force_vector)
    _lib.check(lib.mobi_attention_bwd(C.byref(p), _stream()), "mobi_attention_bwd")
    return dq, dk, dv


# --------------------------------------------------------------------------------------
# matrix-core ops
# --------------------------------------------------------------------------------------
# Split-K launches that finish themselves (mobi_igemm_params.sync: the workgroup that arrives last at an output tile sums the tile's
# slabs, no reduce launch): built, bit-identical to the two-launch form, and SLOWER -- the last block's serial tail (barrier, L2
# atomic, two to five dependent rounds of slab loads, stores) costs 4-6 us more than the fully parallel reduce launch it replaces
# (23.3 -> 27.8 us at 2048 x 1280 x 1280 split 2; profiles/r05_fused_split_lab.txt).  Off; MOBI_FUSED_SPLIT=1 for the A/B.
FUSED_SPLIT = os.environ.get("MOBI_FUSED_SPLIT", "0") == "1"
_SYNC = {}


def _sync_counters(device, nbytes):
    """Arrival counters of the split-K launches that finish themselves: int32 zeros, ONE buffer per (device, stream) -- a
    launch leaves its counters at zero, and the launches of a stream run one after the other, so every launch of the stream
    can use the same ones (two streams running split launches at the same time must not share them)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SYNC.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.zeros((max(nbytes // 4 + 1, 1 << 16),), device=device, dtype=torch.int32)
        _SYNC[key] = buf
    return buf


# A split-K launch's partial sums handed to the GroupNorm that consumes the result (mobi_split_source: the consumer sums the
# slabs while it loads its rows -- bit for bit the reduce launch's arithmetic -- so the producer needs no second launch).
# MOBI_DEFER_SPLIT=0: every split launch finishes itself with its reduce launch (A/B).
DEFER_SPLIT = os.environ.get("MOBI_DEFER_SPLIT", "1") != "0"


class Deferred:
    """The output of a split-K `igemm(..., defer=True)` that has not been summed yet.  `tensor` is the allocated result
    [N,H,W,cout]; it holds the result only after a GroupNorm has consumed the slabs (`groupnorm(x=deferred)` writes it when
    `keep`) or after `finish()`.  Everything that is not a GroupNorm takes `ops.finished(x)`."""

    def __init__(self, tensor, params, ws, refs, keep):
        self.tensor, self.params, self.ws, self.refs, self.keep = tensor, params, ws, refs, keep
        self.count = _lib.load().mobi_igemm_slab_count(C.byref(params))
        assert self.count >= 2, self.count
        self.done = False

    @property
    def shape(self):
        return self.tensor.shape

    @property
    def dtype(self):
        return self.tensor.dtype

    @property
    def device(self):
        return self.tensor.device

    def finish(self):
        """The reduce launch after all (a consumer that is not a GroupNorm of a shape the register form takes)."""
        if not self.done:
            m = self.tensor.numel() // self.tensor.shape[3]
            with _Timed("split_finish", 0.0, self.count * m * self.params.n_packed * 4.0 + self.tensor.numel() * 2.0,
                        f"m={m} n={self.params.n_packed} slabs={self.count}"):
                _lib.check(_lib.load().mobi_igemm_finish(C.byref(self.params), _stream()), "mobi_igemm_finish")
            self._release()
        return self.tensor

    def _release(self):
        self.done = True
        self.ws = self.refs = None


def finished(x):
    """x, or the summed tensor of a Deferred x."""
    return x.finish() if isinstance(x, Deferred) else x


def igemm(x, pw: Packed, *, x2=None, stride=1, pad=None, upsample=False, hout=None, wout=None, rowvec=None,
          rowvec_has_bias=False, residual=None, out=None, out_mode=OUT_ROWS, scale=1.0, weight_per_image=False,
          w_group_stride=0, split_k=None, groups=1, defer=None):
    """x: [N,H,W,C0] (tokens: [N,T,1,C]); x2: optional second source concatenated on channels.
    defer: None, or "keep" / "drop" -- the caller promises that the result's FIRST reader is a GroupNorm over it: a launch that
    splits k then returns a `Deferred` (no reduce launch; "keep": the GroupNorm also writes the summed tensor, "drop": nobody
    else reads it).  A launch that does not split returns its tensor as always.
    rowvec: fp32 [N, cout] added per image; rowvec_has_bias: its producer already added this layer's bias (the
    launch then passes no bias, which keeps it on the register-epilogue kernels).
    groups = g > 1: `pw` holds g stacked matrices [g * cout][k]; image i is multiplied by matrix i // (N / g) (ONE launch for
    the camera images' and the lidar images' projections of a [camera ; lidar] batch); no bias."""
    lib = _lib.load()
    x, x2, residual = finished(x), finished(x2), finished(residual)
    n, hin, win, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[3]
    assert c0 + c1 == pw.cin, (c0, c1, pw.cin)
    if groups > 1:
        assert not weight_per_image and not pw.geglu and pw.bias is None and n % groups == 0 and pw.cout % groups == 0
        pw = Packed(pw.w, None, pw.kh, pw.kw, pw.cin, pw.cout // groups, pw.n_packed // groups, wt=pw.wt)
        w_group_stride = pw.n_packed * pw.kh * pw.kw * pw.cin
    ph, pw_ = (pw.kh // 2, pw.kw // 2) if pad is None else pad
    hl, wl = (hin * 2, win * 2) if upsample else (hin, win)
    if hout is None:
        hout = (hl + 2 * ph - pw.kh) // stride + 1
        wout = (wl + 2 * pw_ - pw.kw) // stride + 1
    if out is None:
        if out_mode == OUT_TRANSPOSED:
            out = torch.empty((n, pw.cout, hout * wout), device=x.device, dtype=x.dtype)
        elif out_mode == OUT_ROWS_F32:
            out = torch.empty((n, hout, wout, pw.cout), device=x.device, dtype=torch.float32)
        else:
            out = torch.empty((n, hout, wout, pw.cout), device=x.device, dtype=x.dtype)
    p = _lib.IgemmParams()
    p.src0, p.src1 = _ptr(x), _ptr(x2)
    p.c0, p.c1, p.batch, p.hin, p.win = c0, c1, n, hin, win
    p.upsample, p.hout, p.wout = int(upsample), hout, wout
    p.kh, p.kw, p.stride, p.pad_h, p.pad_w = pw.kh, pw.kw, stride, ph, pw_
    dense_in = hin * win * c0
    s = _img_stride(x)
    p.src_img_stride = 0 if s == dense_in else s
    p.weight = _ptr(pw.w)
    p.weight_tiled = None if weight_per_image else _ptr(pw.wt)
    p.groups = n if weight_per_image else groups
    p.w_group_stride = w_group_stride
    p.n_packed, p.cout = pw.n_packed, pw.cout
    assert rowvec is not None or not rowvec_has_bias
    p.bias, p.rowvec, p.residual = (None if rowvec_has_bias else _ptr(pw.bias)), _ptr(rowvec), _ptr(residual)
    if rowvec is not None:
        assert rowvec.dtype == torch.float32 and rowvec.stride(1) == 1 and rowvec.shape == (n, pw.cout)
        p.rowvec_stride = rowvec.stride(0)
    dense_out = hout * wout * pw.cout
    if residual is not None:
        s = _img_stride(residual)
        p.res_img_stride = 0 if s == dense_out else s
    s = _img_stride(out)
    p.out = _ptr(out)
    p.out_img_stride = 0 if s == dense_out else s
    p.out_mode = out_mode
    p.epilogue = EPI_GEGLU if pw.geglu else EPI_NONE
    p.scale = scale
    p.k_order = pw.k_order
    if pw.svec is not None:                                  # LayerNorm folded into this launch: never split (a block sweeps all of k)
        assert pw.svec.dtype == torch.float32 and pw.svec.numel() == pw.n_packed and pw.bias is not None and not rowvec_has_bias
        p.ln_svec, p.ln_eps = _ptr(pw.svec), pw.ln_eps
        split_k = 1
    if pw.k_order and (c0 % 64 or (c1 and c1 % 64)):
        raise ValueError("chunk-major weights need 64-channel-aligned sources")
    p.dtype = _dt(x.dtype)
    splits = split_k if split_k is not None else lib.mobi_igemm_plan_splits(C.byref(p))
    if splits > 1:
        ws = torch.empty(lib.mobi_igemm_workspace_bytes(C.byref(p), splits), device=x.device, dtype=torch.uint8)
        p.split_k, p.ws = splits, _ptr(ws)
        if FUSED_SPLIT:
            p.sync = _ptr(_sync_counters(x.device, lib.mobi_igemm_sync_bytes(C.byref(p), splits)))
    flops = 2.0 * n * hout * wout * (pw.n_packed if pw.geglu else pw.cout) * pw.kh * pw.kw * pw.cin
    nbytes = (x.numel() + (0 if x2 is None else x2.numel())) * 2 + pw.w.numel() * 2 * (n if weight_per_image else 1) \
        + out.numel() * out.element_size() + (0 if residual is None else residual.numel() * 2)
    tag = ""
    if _PROFILE is not None:
        kern = ("staged128", "staged256", "direct_lds", "pingpong", "ring128", "ring256", "ring128w", "small")[lib.mobi_igemm_kernel_variant(C.byref(p))]
        if pw.svec is not None:
            kern += "_ln"                                    # the LayerNorm-folded instantiation (another kernel symbol)
        tag = f"kern={kern} m={n * hout * wout} n={pw.n_packed} k={pw.kh * pw.kw * pw.cin} tap={pw.kh}x{pw.kw} " \
              f"split={splits} mode={out_mode}"
    deferred = (defer is not None and DEFER_SPLIT and splits > 1 and not FUSED_SPLIT and out_mode == OUT_ROWS and out.is_contiguous()
                and (residual is None or residual.is_contiguous()))
    if deferred:
        p.defer_finish = 1
    with _Timed("igemm", flops, nbytes, tag):
        _lib.check(lib.mobi_igemm(C.byref(p), _stream()), "mobi_igemm")
    if deferred:
        # (`refs`: the operands the finishing launch reads must outlive this call)
        return Deferred(out, p, ws, (pw, rowvec, residual, x, x2), keep=defer == "keep")
    return out


def linear(x, pw: Packed, **kw):
    """x: [N, T, C] tokens -> [N, T, cout]."""
    n, t, c = x.shape
    out = kw.pop("out", None)
    res = kw.pop("residual", None)
    if out is not None and out.dim() == 3:
        out = out.unsqueeze(2) if kw.get("out_mode", OUT_ROWS) != OUT_TRANSPOSED else out
    if res is not None:
        res = res.unsqueeze(2)
    y = igemm(x.unsqueeze(2), pw, out=out, residual=res, **kw)
    return y if y.dim() == 3 else y.squeeze(2)


GN_OUT_T, GN_OUT_SPLIT, GN_OUT_F32, GN_OUT_SPLIT3 = 0, 1, 2, 3


def groupnorm(x, gamma, beta, eps, silu, x2=None, out_mode=GN_OUT_T, dtype=None):
    """x: T or fp32 (one source) [N,H,W,C].  out_mode GN_OUT_SPLIT: T [N,H,W,2C] = hi | lo (hi = T(y), lo = T(y - hi)) for a
    consumer whose weights are duplicated along its input channels; GN_OUT_SPLIT3: T [N,H,W,3C] = hi | lo | hi for weights
    [W ; W ; W - T(W)] (Conv2d.packed_split: the weights' rounding corrected too); GN_OUT_F32: fp32 [N,H,W,C].  dtype: the storage type T when
    x is fp32."""
    lib = _lib.load()
    x2 = finished(x2)
    split = None
    if isinstance(x, Deferred):
        n, h, w, c0 = x.shape
        if (x.done or out_mode != GN_OUT_T or dtype not in (None, x.dtype)
                or not lib.mobi_groupnorm_takes_split(c0, 0 if x2 is None else x2.shape[3], n, h * w)):
            x = x.finish()
        else:
            split, x = x, x.tensor
    n, h, w, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[3]
    assert x.is_contiguous() and (x2 is None or x2.is_contiguous())
    src_f32 = x.dtype == torch.float32
    t = dtype if dtype is not None else x.dtype
    assert t in (torch.float16, torch.bfloat16) and not (src_f32 and x2 is not None)
    oc = (c0 + c1) * {GN_OUT_SPLIT: 2, GN_OUT_SPLIT3: 3}.get(out_mode, 1)
    out = torch.empty((n, h, w, oc), device=x.device, dtype=torch.float32 if out_mode == GN_OUT_F32 else t)
    ws = torch.empty(lib.mobi_groupnorm_workspace_bytes(n, h * w), device=x.device, dtype=torch.uint8)
    p = _lib.GroupNormParams()
    p.src0, p.src1, p.c0, p.c1, p.batch, p.hw = _ptr(x), _ptr(x2), c0, c1, n, h * w
    if split is not None:
        q = split.params
        ss = _lib.SplitSource()
        ss.slabs, ss.count, ss.row_stride = q.ws, split.count, q.n_packed
        ss.bias, ss.rowvec, ss.rowvec_stride = q.bias, q.rowvec, q.rowvec_stride
        ss.residual, ss.res_img_stride = q.residual, q.res_img_stride
        ss.finished = _ptr(x) if split.keep else None
        p.src0 = None
        p.src0_split = C.pointer(ss)
    p.gamma, p.beta, p.eps, p.silu = _ptr(gamma), _ptr(beta), eps, int(silu)
    p.out, p.ws, p.dtype = _ptr(out), _ptr(ws), _dt(t)
    p.src_f32, p.out_mode = int(src_f32), out_mode
    p.sync = _ptr(_sync_counters(x.device, 4 * n))
    # algorithmic bytes: the tensor read once and written once (2 B per element each way)
    tag = f"n={n} hw={h * w} c={c0 + c1}"
    nbytes = 2.0 * out.numel() * 2
    if split is not None:
        tag += f" slabs={split.count}"
        nbytes += split.count * n * h * w * split.params.n_packed * 4.0 - c0 * n * h * w * 2.0 * (0 if split.keep else 1)
    with _Timed("groupnorm", 0.0, nbytes, tag):
        _lib.check(lib.mobi_groupnorm(C.byref(p), _stream()), "mobi_groupnorm")
    if split is not None:
        split._release()
        if not split.keep:
            split.tensor = None                              # (never written: nobody may read it)
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    """x: [N, T, C], possibly a batch-strided view; returns a dense tensor."""
    lib = _lib.load()
    n, t, c = x.shape
    out = torch.empty((n, t, c), device=x.device, dtype=x.dtype)
    p = _lib.LayerNormParams()
    p.src, p.out, p.images, p.rows_per_image, p.channels = _ptr(x), _ptr(out), n, t, c
    s = _img_stride(x)
    p.src_img_stride = 0 if s == t * c else s
    p.out_img_stride = 0
    p.gamma, p.beta, p.eps, p.dtype = _ptr(gamma), _ptr(beta), eps, _dt(x.dtype)
    with _Timed("layernorm", 0.0, 2.0 * out.numel() * 2):
        _lib.check(lib.mobi_layernorm(C.byref(p), _stream()), "mobi_layernorm")
    return out


LOG2E = 1.4426950408889634


def attention(q, k, v, heads, scale, v_rows=False, q_log2_scaled=False):
    """q: [N, Tq, >=C] view (row stride may exceed C: stacked projections), k: [N, Tk, ...];
    v_rows=False: v is V^T [N, C, Tk];  v_rows=True: v is [N, Tk, >=C] row-major like k.  -> out [N, Tq, C].
    q_log2_scaled: q already carries scale * log2(e) (folded into the packed to_q weights); `scale` is then unused."""
    lib = _lib.load()
    n, tq = q.shape[0], q.shape[1]
    tk = k.shape[1]
    c = v.shape[2] if v_rows else v.shape[1]
    dh = c // heads
    out = torch.empty((n, tq, c), device=q.device, dtype=q.dtype)
    p = _lib.AttentionParams()
    p.q, p.q_img_stride, p.q_row_stride = _ptr(q), q.stride(0), q.stride(1)
    p.k, p.k_img_stride, p.k_row_stride = _ptr(k), k.stride(0), k.stride(1)
    p.vt, p.vt_img_stride, p.vt_row_stride = _ptr(v), v.stride(0), v.stride(1)
    p.v_layout = 1 if v_rows else 0
    p.q_log2_scaled = 1 if q_log2_scaled else 0
    p.out, p.out_img_stride, p.out_row_stride = _ptr(out), out.stride(0), out.stride(1)
    for t_ in (q, k, v):
        assert t_.shape[2] == 1 or t_.stride(2) == 1
    p.images, p.heads, p.dh, p.tq, p.tk, p.scale, p.dtype = n, heads, dh, tq, tk, scale, _dt(q.dtype)
    with _Timed("attention", 4.0 * n * heads * tq * tk * dh, 0.0, f"n={n} heads={heads} tq={tq} tk={tk} dh={dh}"):
        _lib.check(lib.mobi_attention(C.byref(p), _stream()), "mobi_attention")
    return out


def ctx_attention(q, k, v, heads, scale):
    """q: [N, T, C] dense T; k, v: [N, tk, C] fp32."""
    lib = _lib.load()
    n, t, c = q.shape
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    out = torch.empty_like(q)
    p = _lib.CtxAttentionParams()
    p.q, p.out, p.k, p.v = _ptr(q), _ptr(out), _ptr(k), _ptr(v)
    p.images, p.heads, p.dh, p.tq, p.tk, p.scale, p.dtype = n, heads, c // heads, t, k.shape[1], scale, _dt(q.dtype)
    _lib.check(lib.mobi_ctx_attention(C.byref(p), _stream()), "mobi_ctx_attention")
    return out


def two_key_adapter_fuses_ln(channels, total_rows):
    """Whether mobi_two_key_adapter writes the pair of LayerNorm results (`ln_pair`) at this width and row count."""
    return bool(_lib.load().mobi_two_key_adapter_fuses_ln(int(channels), int(total_rows)))


def two_key_adapter(x, a, a_sum, c, u, b, eps, out=None, ln_pair=None):
    """x: [N, T, C] tokens (T storage type, possibly a batch-strided view); a, u: fp32 [N, H, C]; a_sum, c: fp32 [N, H];
    b: fp32 [N, C].  Returns x + b + sum_h sigmoid(rstd * (x . a_h - mean * a_sum_h) + c_h) * u_h with the token's
    LayerNorm statistics (include/mobi_engine.h, mobi_two_key_adapter); out=x updates in place.
    ln_pair = ((gamma0, beta0), (gamma1, beta1), eps): also returns (LN_0 of the even images, LN_1 of the odd images) of
    the result, [N / 2, T, C] each -- what the cross-modal step normalises next (two_key_adapter_fuses_ln(C, N * T) must hold)."""
    lib = _lib.load()
    n, t, ch = x.shape
    if out is None:
        out = torch.empty((n, t, ch), device=x.device, dtype=x.dtype)
    for tns in (a, a_sum, c, u, b):
        assert tns.dtype == torch.float32 and tns.is_contiguous()
    assert a.shape == u.shape == (n, a.shape[1], ch) and a_sum.shape == c.shape == (n, a.shape[1]) and b.shape == (n, ch)
    p = _lib.TwoKeyAdapterParams()
    p.x, p.out = _ptr(x), _ptr(out)
    sx, so = _img_stride(x), _img_stride(out)
    p.x_img_stride = 0 if sx == t * ch else sx
    p.out_img_stride = 0 if so == t * ch else so
    p.a, p.a_sum, p.c, p.u, p.b = _ptr(a), _ptr(a_sum), _ptr(c), _ptr(u), _ptr(b)
    p.images, p.rows_per_image, p.channels, p.heads, p.eps, p.dtype = n, t, ch, a.shape[1], eps, _dt(x.dtype)
    lns = None
    nbytes = 2.0 * x.numel() * 2
    if ln_pair is not None:
        (g0, b0), (g1, b1), ln_eps = ln_pair
        assert n % 2 == 0
        for tns in (g0, b0, g1, b1):
            assert tns.dtype == torch.float32 and tns.is_contiguous() and tns.numel() == ch
        # (halves of ONE buffer [camera images ; lidar images]: the cross-modal step multiplies both by their own to_q in one
        #  grouped launch, mobi_igemm_params.groups = 2)
        both = torch.empty((n, t, ch), device=x.device, dtype=x.dtype)
        lns = (both[:n // 2], both[n // 2:])
        for i, (tn, g, bb) in enumerate(((lns[0], g0, b0), (lns[1], g1, b1))):
            p.ln_out[i], p.ln_gamma[i], p.ln_beta[i] = _ptr(tn), _ptr(g), _ptr(bb)
        p.ln_eps = ln_eps
        nbytes += x.numel() * 2
    with _Timed("two_key_adapter", 0.0, nbytes):
        _lib.check(lib.mobi_two_key_adapter(C.byref(p), _stream()), "mobi_two_key_adapter")
    return out if lns is None else (out, lns)


def softmax_rows(s, dtype):
    lib = _lib.load()
    assert s.dtype == torch.float32 and s.is_contiguous()
    out = torch.empty(s.shape, device=s.device, dtype=dtype)
    rows = s.numel() // s.shape[-1]
    _lib.check(lib.mobi_softmax_rows(_ptr(s), _ptr(out), rows, s.shape[-1], _dt(dtype), _stream()), "mobi_softmax_rows")
    return out


# --------------------------------------------------------------------------------------
# small ops
# --------------------------------------------------------------------------------------
def skinny_linear(x, w, bias=None, pre_act=ACT_NONE, post_act=ACT_NONE, out=None):
    """x: fp32 [m, k] (row stride free), w: T [n, k] -> fp32 [m, n]."""
    lib = _lib.load()
    m, k = x.shape
    n = w.shape[0]
    assert x.dtype == torch.float32 and x.stride(1) == 1 and w.is_contiguous()
    if out is None:
        out = torch.empty((m, n), device=x.device, dtype=torch.float32)
    for m0 in range(0, m, 16):                      # one launch per 16 rows (the kernel's LDS row block)
        xs, os_ = x[m0:m0 + 16], out[m0:m0 + 16]
        p = _lib.SkinnyLinearParams()
        p.x, p.m, p.k, p.x_row_stride = _ptr(xs), xs.shape[0], k, x.stride(0)
        p.weight, p.bias, p.out, p.n, p.out_row_stride = _ptr(w), _ptr(bias), _ptr(os_), n, out.stride(0)
        p.pre_act, p.post_act, p.dtype = pre_act, post_act, _dt(w.dtype)
        _lib.check(lib.mobi_skinny_linear(C.byref(p), _stream()), "mobi_skinny_linear")
    return out


def linear_f32(x, w, bias=None):
    """fp32 [m, k] x fp32 [n, k]^T (+ fp32 bias [n]) -> fp32 [m, n]; plain fp32 FMA chains (mobi_linear_f32)."""
    lib = _lib.load()
    assert x.dtype == w.dtype == torch.float32 and x.dim() == w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    assert x.shape[1] == w.shape[1] and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous()))
    _dev(x), _dev(w)
    out = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=torch.float32)
    _lib.check(lib.mobi_linear_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), x.shape[0], w.shape[0], x.shape[1], x.stride(0),
                                   w.stride(0), out.stride(0), _stream()), "mobi_linear_f32")
    return out


def layernorm_rows_f32(x, gamma, beta, eps=1e-5):
    """fp32 [rows, cols] (row stride free) -> fp32 [rows, cols]."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    out = torch.empty((x.shape[0], x.shape[1]), device=x.device, dtype=torch.float32)
    _lib.check(lib.mobi_layernorm_rows_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), x.shape[0], x.shape[1], x.stride(0),
                                           out.stride(0), eps, _stream()), "mobi_layernorm_rows_f32")
    return out


def quick_gelu(x):
    """T tensor -> x * sigmoid(1.702 x), same shape."""
    lib = _lib.load()
    assert x.is_contiguous() and x.numel() % 8 == 0
    out = torch.empty_like(x)
    _lib.check(lib.mobi_quick_gelu(_ptr(x), _ptr(out), x.numel(), _dt(x.dtype), _stream()), "mobi_quick_gelu")
    return out


def timestep_embedding(t, freqs):
    lib = _lib.load()
    n, half = t.shape[0], freqs.shape[0]
    assert t.dtype == torch.int64 and freqs.dtype == torch.float32
    out = torch.empty((n, 2 * half), device=t.device, dtype=torch.float32)
    _lib.check(lib.mobi_timestep_embedding(_ptr(t), _ptr(freqs), _ptr(out), n, half, _stream()),
               "mobi_timestep_embedding")
    return out


def conv_small_cin(srcs, weight, bias, kh, kw, pad, dtype, out_f32_nchw=False):
    """srcs: list of fp32 NCHW tensors (<= 3) concatenated on channels; weight fp32 [cout, cin*kh*kw]."""
    lib = _lib.load()
    n, _, h, w = srcs[0].shape
    cout = weight.shape[0]
    p = _lib.ConvSmallCinParams()
    for i, s in enumerate(srcs):
        assert s.dtype == torch.float32 and s.is_contiguous() and s.shape[0] == n and s.shape[2:] == (h, w)
        p.src[i], p.c[i] = s.data_ptr(), s.shape[1]
        _dev(s)
    p.batch, p.h, p.w, p.kh, p.kw, p.pad_h, p.pad_w = n, h, w, kh, kw, pad[0], pad[1]
    p.weight, p.bias, p.cout = _ptr(weight), _ptr(bias), cout
    if out_f32_nchw:
        out = torch.empty((n, cout, h, w), device=srcs[0].device, dtype=torch.float32)
    else:
        out = torch.empty((n, h, w, cout), device=srcs[0].device, dtype=dtype)
    p.out, p.out_f32_nchw, p.dtype = _ptr(out), int(out_f32_nchw), _dt(dtype)
    _lib.check(lib.mobi_conv_small_cin(C.byref(p), _stream()), "mobi_conv_small_cin")
    return out


def conv_small_cout(x, pw: Packed, pad=None, clamp=None):
    """x: T [N,H,W,Cin] -> fp32 NCHW [N,cout,H,W]."""
    lib = _lib.load()
    n, h, w, cin = x.shape
    assert x.is_contiguous() and cin == pw.cin and pw.k_order == 0
    out = torch.empty((n, pw.cout, h, w), device=x.device, dtype=torch.float32)
    p = _lib.ConvSmallCoutParams()
    ph, pw_ = (pw.kh // 2, pw.kw // 2) if pad is None else pad
    p.src, p.cin, p.batch, p.h, p.w = _ptr(x), cin, n, h, w
    p.kh, p.kw, p.pad_h, p.pad_w = pw.kh, pw.kw, ph, pw_
    p.weight, p.bias, p.cout, p.out = _ptr(pw.w), _ptr(pw.bias), pw.cout, _ptr(out)
    if clamp is not None:
        p.clamp, p.clamp_lo, p.clamp_hi = 1, clamp[0], clamp[1]
    p.in_scale, p.dtype = 1.0, _dt(x.dtype)
    _lib.check(lib.mobi_conv_small_cout(C.byref(p), _stream()), "mobi_conv_small_cout")
    return out


def pack_sources(srcs, dtype, c_pad=32):
    """fp32 NCHW sources (<= 3), concatenated on channels -> T [N,H,W,c_pad] (zero padded)."""
    lib = _lib.load()
    n, _, h, w = srcs[0].shape
    for s_ in srcs:
        assert s_.dtype == torch.float32 and s_.is_contiguous() and s_.shape[0] == n and s_.shape[2:] == (h, w)
    ss = list(srcs) + [None] * (3 - len(srcs))
    cs = [0 if s_ is None else s_.shape[1] for s_ in ss]
    out = torch.empty((n, h, w, c_pad), device=srcs[0].device, dtype=dtype)
    _lib.check(lib.mobi_pack_nchw_sources(_ptr(ss[0]), _ptr(ss[1]), _ptr(ss[2]), cs[0], cs[1], cs[2], n, h * w,
                                          c_pad, _ptr(out), _dt(dtype), _stream()), "mobi_pack_nchw_sources")
    return out


def to_nhwc(x, dtype):
    """fp32 NCHW -> T NHWC"""
    lib = _lib.load()
    n, c, h, w = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty((n, h, w, c), device=x.device, dtype=dtype)
    _lib.check(lib.mobi_nchw_f32_to_nhwc(_ptr(x), _ptr(out), n, c, h * w, _dt(dtype), _stream()), "nchw_to_nhwc")
    return out


def to_nchw_f32(x):
    lib = _lib.load()
    n, h, w, c = x.shape
    assert x.is_contiguous()
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    _lib.check(lib.mobi_nhwc_to_nchw_f32(_ptr(x), _ptr(out), n, c, h * w, _dt(x.dtype), _stream()), "nhwc_to_nchw")
    return out


# --------------------------------------------------------------------------------------
# sampler / latent arithmetic (fp32 NCHW)
# --------------------------------------------------------------------------------------
def ddim_step(x, e_cond, *, e_uncond=None, noise=None, cfg_scale=1.0, a_t=1.0, a_prev=1.0, sigma_t=0.0,
              sqrt_one_minus_at=0.0, temperature=1.0, want_e=False, coef_dev=None):
    """coef_dev: fp32 device tensor {a_t, a_prev, sigma_t, sqrt_one_minus_at} replacing the by-value coefficients
    (graph-captured steps, mobi_amd/graph.py)."""
    lib = _lib.load()
    for t in (x, e_cond, e_uncond, noise):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    x_prev, pred = torch.empty_like(x), torch.empty_like(x)
    e_out = torch.empty_like(x) if want_e else None
    p = _lib.DdimStepParams()
    p.x, p.e_cond, p.e_uncond, p.noise = _ptr(x), _ptr(e_cond), _ptr(e_uncond), _ptr(noise)
    p.x_prev, p.pred_x0, p.e_out, p.n = _ptr(x_prev), _ptr(pred), _ptr(e_out), x.numel()
    p.cfg_scale, p.a_t, p.a_prev, p.sigma_t = cfg_scale, a_t, a_prev, sigma_t
    p.sqrt_one_minus_at, p.temperature = sqrt_one_minus_at, temperature
    if coef_dev is not None:
        assert coef_dev.dtype == torch.float32 and coef_dev.numel() == 4 and coef_dev.is_contiguous()
        p.coef_dev = _ptr(coef_dev)
    _lib.check(lib.mobi_ddim_step(C.byref(p), _stream()), "mobi_ddim_step")
    return x_prev, pred, e_out


def lincomb4(es, cs):
    lib = _lib.load()
    es = list(es) + [None] * (4 - len(es))
    cs = list(cs) + [0.0] * (4 - len(cs))
    out = torch.empty_like(es[0])
    _lib.check(lib.mobi_lincomb4(_ptr(out), _ptr(es[0]), _ptr(es[1]), _ptr(es[2]), _ptr(es[3]),
                                 cs[0], cs[1], cs[2], cs[3], out.numel(), _stream()), "mobi_lincomb4")
    return out


def q_sample(x0, noise, t, sqrt_ac, sqrt_1m_ac):
    """fp32 [B, ...] x0 / noise, int64 [B] t and the two fp32 schedule tables, all on the device."""
    lib = _lib.load()
    for t_ in (x0, noise, sqrt_ac, sqrt_1m_ac):
        assert t_.dtype == torch.float32 and t_.is_contiguous()
    assert t.dtype == torch.int64 and t.is_contiguous() and t.numel() == x0.shape[0] and x0.shape == noise.shape
    out = torch.empty_like(x0)
    _lib.check(lib.mobi_q_sample(_ptr(x0), _ptr(noise), _ptr(t), _ptr(sqrt_ac), _ptr(sqrt_1m_ac), _ptr(out),
                                 x0.shape[0], x0[0].numel(), sqrt_ac.numel(), _stream()), "mobi_q_sample")
    return out


def mask_blend_(img, x0, noise, mask, sqrt_ac_t, sqrt_1m_ac_t):
    lib = _lib.load()
    b, c, h, w = img.shape
    _lib.check(lib.mobi_mask_blend(_ptr(img), _ptr(x0), _ptr(noise), _ptr(mask), sqrt_ac_t, sqrt_1m_ac_t,
                                   b, c, h * w, _stream()), "mobi_mask_blend")
    return img


def range_denorm(sample, min_d=None, max_d=None, alpha=0.75, object_norm=True, int_norm=True):
    """sample: fp32 [B, 2, H, W] clamped lidar range sample -> (depth [B,1,H,W], intensity [B,1,H,W]) de-normalised as
    log_data does on the host (include/mobi_engine.h, mobi_range_denorm).  min_d / max_d: fp32 [B] on the device."""
    lib = _lib.load()
    assert sample.dtype == torch.float32 and sample.is_contiguous() and sample.shape[1] == 2
    b, _, h, w = sample.shape
    depth = torch.empty((b, 1, h, w), device=sample.device, dtype=torch.float32)
    inten = torch.empty((b, 1, h, w), device=sample.device, dtype=torch.float32)
    if object_norm:
        assert min_d.dtype == max_d.dtype == torch.float32 and min_d.numel() == max_d.numel() == b
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float64).to(torch.float32))      # as torch wraps a Python scalar
    _lib.check(lib.mobi_range_denorm(_ptr(sample), _ptr(min_d), _ptr(max_d), f32(alpha), f32(2 * alpha), f32(alpha - 1),
                                     f32(1 - alpha), int(object_norm), int(int_norm), _ptr(depth), _ptr(inten), b, h * w,
                                     _stream()), "mobi_range_denorm")
    return depth, inten


def posterior_sample(moments, noise, out, c_off, scale):
    lib = _lib.load()
    b, c2, h, w = moments.shape
    c = c2 // 2
    _lib.check(lib.mobi_posterior_sample(_ptr(moments), _ptr(noise), _ptr(out), b, c, h * w, out.shape[1], c_off,
                                         scale, _stream()), "mobi_posterior_sample")
    return out


def nearest_resize(src, hout, wout, out=None, c_off=0):
    """fp32 [B, C, H, W] -> writes planes [B, c_off:c_off+C] of `out` ([B, Ct, hout, wout])."""
    lib = _lib.load()
    b, c, h, w = src.shape
    if out is None:
        out = torch.empty((b, c, hout, wout), device=src.device, dtype=torch.float32)
    ct = out.shape[1]
    if ct == c:
        _lib.check(lib.mobi_nearest_resize(_ptr(src), _ptr(out), b * c, h, w, hout, wout, hout * wout, _stream()),
                   "mobi_nearest_resize")
    else:
        assert c == 1
        view = out[:, c_off]
        _lib.check(lib.mobi_nearest_resize(_ptr(src), C.c_void_p(view.data_ptr()), b, h, w, hout, wout,
                                           ct * hout * wout, _stream()), "mobi_nearest_resize")
    return out


# --------------------------------------------------------------------------------------
# harness post-processing
# --------------------------------------------------------------------------------------
def _i32(t, device):
    return torch.as_tensor(t).to(device=device, dtype=torch.int32).reshape(-1).contiguous()


def range_paste(sample_depth, depth_orig, crop_left, width_crop, sample_int=None, int_orig=None, pitch=None, yaw=None,
                planes=None, gt_mask=None, depth_interval=(1.4, 54.0)):
    """sample_*: fp32 [B, Hc, Wc] (de-normalised range sample), *_orig: fp32 [B, H0, W0]; crop_left / width_crop: [B] ints.
    Returns dict(depth_unc, int_unc[, depth_final, int_final, pred_mask]) -- include/mobi_engine.h, mobi_range_paste."""
    lib = _lib.load()
    dev = sample_depth.device
    b, hc, wc = sample_depth.shape
    _, h0, w0 = depth_orig.shape
    f = lambda t: None if t is None else _dev(t).to(torch.float32).contiguous()
    sample_depth, sample_int, depth_orig, int_orig, pitch, yaw, planes = map(
        f, (sample_depth, sample_int, depth_orig, int_orig, pitch, yaw, planes))
    cl, wcr = _i32(crop_left, dev), _i32(width_crop, dev)
    assert cl.numel() == wcr.numel() == b and depth_orig.shape[0] == b
    out = {"depth_unc": torch.empty_like(depth_orig)}
    if sample_int is not None:
        out["int_unc"] = torch.empty_like(depth_orig)
    gm = None
    if planes is not None:
        assert planes.shape == (b, 6, 4) and pitch.shape == yaw.shape == depth_orig.shape
        out["depth_final"] = torch.empty_like(depth_orig)
        if sample_int is not None:
            out["int_final"] = torch.empty_like(depth_orig)
        out["pred_mask"] = torch.empty(depth_orig.shape, device=dev, dtype=torch.uint8)
        gm = None if gt_mask is None else gt_mask.to(device=dev, dtype=torch.uint8).contiguous()
    p = _lib.RangePasteParams()
    p.sample_depth, p.sample_int, p.depth_orig, p.int_orig = _ptr(sample_depth), _ptr(sample_int), _ptr(depth_orig), _ptr(int_orig)
    p.pitch, p.yaw, p.gt_mask, p.planes = _ptr(pitch), _ptr(yaw), _ptr(gm), _ptr(planes)
    p.crop_left, p.width_crop = _ptr(cl), _ptr(wcr)
    p.depth_unc, p.int_unc = _ptr(out["depth_unc"]), _ptr(out.get("int_unc"))
    p.depth_final, p.int_final, p.pred_mask = _ptr(out.get("depth_final")), _ptr(out.get("int_final")), _ptr(out.get("pred_mask"))
    p.batch, p.hc, p.wc, p.h0, p.w0 = b, hc, wc, h0, w0
    p.depth_min, p.depth_max = float(depth_interval[0]), float(depth_interval[1])
    _lib.check(lib.mobi_range_paste(C.byref(p), _stream()), "mobi_range_paste")
    return out


def range_prepare(depth_orig, int_orig, inst_orig, crop_left, width_crop, min_depth, max_depth, edit_mask, *, height, width,
                  alpha=0.75, object_norm=True, int_norm=False):
    """Dataset side, a whole batch in one launch (include/mobi_engine.h, mobi_range_prepare): sweeps fp32 [B, H0, W0]
    (depth code, raw intensity 0..255, instance mask), crop windows [B], object depth ranges [B], edit masks
    [B, 1, height, width] -> (range_data [B, 2, h, w], range_data_inpaint [B, 2, h, w], range_instance_mask [B, 1, h, w])."""
    lib = _lib.load()
    dev = depth_orig.device
    b, h0, w0 = depth_orig.shape
    f = lambda t: None if t is None else _dev(t).to(torch.float32).contiguous()
    depth_orig, int_orig, inst_orig, min_depth, max_depth, edit_mask = map(
        f, (depth_orig, int_orig, inst_orig, min_depth, max_depth, edit_mask))
    wc_given = torch.as_tensor(width_crop)
    if not wc_given.is_cuda and (int(wc_given.max()) > width or int(wc_given.min()) <= 0):   # (device-resident windows: no
        raise ValueError("range_prepare: crop windows must be 1 ... width columns wide")   #  read-back; the kernel clamps)
    cl, wcr = _i32(crop_left, dev), _i32(width_crop, dev)
    assert cl.numel() == wcr.numel() == b and edit_mask.numel() == b * height * width
    rd = torch.empty((b, 2, height, width), device=dev, dtype=torch.float32)
    rdi = torch.empty_like(rd)
    inst = None if inst_orig is None else torch.empty((b, 1, height, width), device=dev, dtype=torch.float32)
    p = _lib.RangePrepareParams()
    p.depth_orig, p.int_orig, p.inst_orig = _ptr(depth_orig), _ptr(int_orig), _ptr(inst_orig)
    p.crop_left, p.width_crop, p.min_depth, p.max_depth = _ptr(cl), _ptr(wcr), _ptr(min_depth), _ptr(max_depth)
    p.edit_mask, p.range_data, p.range_data_inpaint, p.inst_out = _ptr(edit_mask), _ptr(rd), _ptr(rdi), _ptr(inst)
    p.batch, p.h0, p.w0, p.height, p.width = b, h0, w0, height, width
    p.alpha, p.object_norm, p.int_norm = float(alpha), int(object_norm), int(int_norm)
    _lib.check(lib.mobi_range_prepare(C.byref(p), _stream()), "mobi_range_prepare")
    return rd, rdi, inst


def box_mask(corners_xy, height, width, *, want_mask=True, want_stats=False):
    """[B, 8, 2] projected box corners (pixels; truncated towards zero like `.astype(np.int32)`) -> fp32 [B, height, width]
    edit masks (0 = edit region) and / or int32 [B, 5] = (edit pixels, min x, max x, min y, max y)."""
    lib = _lib.load()
    c = _dev(corners_xy)
    c = (c.to(torch.float64).trunc() if c.is_floating_point() else c).to(torch.int32).contiguous()
    assert c.dim() == 3 and c.shape[1:] == (8, 2)
    b = c.shape[0]
    out = torch.empty((b, height, width), device=c.device, dtype=torch.float32) if want_mask else None
    stats = None
    if want_stats:
        stats = torch.zeros((b, 5), device=c.device, dtype=torch.int32)
        stats[:, 1], stats[:, 2], stats[:, 3], stats[:, 4] = width, -1, height, -1
    _lib.check(lib.mobi_box_mask(_ptr(c), _ptr(out), _ptr(stats), b, height, width, _stream()), "mobi_box_mask")
    if want_mask and want_stats:
        return out, stats
    return out if want_mask else stats


def image_prepare(frames, corners_xy, invert, crop, *, height, width):
    """Dataset side, the camera view of a batch (include/mobi_engine.h, mobi_image_prepare): frames uint8 [B, H, W, 3],
    int corners [B, 8, 2], invert [B], crop [B, 4] = (left, top, crop_w, crop_h) -> (GT [B, 3, h, w], inpaint_image,
    inpaint_mask [B, 1, h, w])."""
    lib = _lib.load()
    frames = _dev(frames).contiguous()
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
    b, H, W, _ = frames.shape
    dev = frames.device
    c = torch.as_tensor(corners_xy).to(dev)
    c = (c.to(torch.float64).trunc() if c.is_floating_point() else c).to(torch.int32).contiguous()
    inv, cr = _i32(invert, dev), _i32(crop, dev)
    assert c.shape == (b, 8, 2) and inv.numel() == b and cr.numel() == 4 * b
    crh = torch.as_tensor(crop)
    if not crh.is_cuda:
        crh = crh.reshape(b, 4)
        if bool(((crh[:, 0] < 0) | (crh[:, 1] < 0) | (crh[:, 2] <= 0) | (crh[:, 3] <= 0) | (crh[:, 0] + crh[:, 2] > W) |
                 (crh[:, 1] + crh[:, 3] > H)).any()):
            raise ValueError("image_prepare: a crop window leaves its frame")
    gt = torch.empty((b, 3, height, width), device=dev, dtype=torch.float32)
    inp, mask = torch.empty_like(gt), torch.empty((b, 1, height, width), device=dev, dtype=torch.float32)
    p = _lib.ImagePrepareParams()
    p.frames, p.corners_xy, p.invert, p.crop = _ptr(frames), _ptr(c), _ptr(inv), _ptr(cr)
    p.gt, p.inpaint, p.mask = _ptr(gt), _ptr(inp), _ptr(mask)
    p.batch, p.H, p.W, p.height, p.width = b, H, W, height, width
    _lib.check(lib.mobi_image_prepare(C.byref(p), _stream()), "mobi_image_prepare")
    return gt, inp, mask


def lidar_metrics(pred, gt, inst_mask, box_mask, width_crop, pool_h=32):
    """fp32 [B, H, W] each (0/1 masks) -> fp32 [B, 2 regions (object, mask), 3 (rmse, median, count)] on the device."""
    lib = _lib.load()
    dev = pred.device
    b, h, w = pred.shape
    f = lambda t: _dev(t).to(torch.float32).reshape(b, h, w).contiguous()
    pred, gt, inst_mask, box_mask = map(f, (pred, gt, inst_mask, box_mask))
    wcr = _i32(width_crop, dev)
    out = torch.empty((b, 2, 3), device=dev, dtype=torch.float32)
    p = _lib.LidarMetricsParams()
    p.pred, p.gt, p.inst_mask, p.box_mask, p.width_crop, p.out = _ptr(pred), _ptr(gt), _ptr(inst_mask), _ptr(box_mask), _ptr(wcr), _ptr(out)
    p.batch, p.h, p.w, p.pool_h, p.max_width = b, h, w, pool_h, w       # width_crop <= w: sort space for the widest case
    _lib.check(lib.mobi_lidar_metrics(C.byref(p), _stream()), "mobi_lidar_metrics")
    return out


def paste_patch(patch, frame, top, left, crop_h, crop_w):
    """patch: fp32 [3, hs, ws] RGB in [-1, 1]; frame: uint8 [H, W, 3] BGR (written in place)."""
    lib = _lib.load()
    assert patch.dtype == torch.float32 and patch.is_contiguous() and frame.dtype == torch.uint8 and frame.is_contiguous()
    _lib.check(lib.mobi_paste_patch(_ptr(patch), patch.shape[1], patch.shape[2], _ptr(frame), frame.shape[0], frame.shape[1],
                                    int(top), int(left), int(crop_h), int(crop_w), _stream()), "mobi_paste_patch")
    return frame


def gaussian_blur(src, kern):
    """src: fp32 [H, W]; kern: fp32 [ksize] taps; BORDER_REFLECT_101."""
    lib = _lib.load()
    assert src.dtype == torch.float32 and src.is_contiguous() and kern.dtype == torch.float32 and kern.is_contiguous()
    tmp, dst = torch.empty_like(src), torch.empty_like(src)
    _lib.check(lib.mobi_gaussian_blur(_ptr(src), _ptr(tmp), _ptr(dst), src.shape[0], src.shape[1], _ptr(kern), kern.numel(),
                                      _stream()), "mobi_gaussian_blur")
    return dst


def blend_frame(mask_blur, image, pred):
    """mask_blur fp32 [H, W], image fp32 [3, H, W] RGB in [-1, 1], pred uint8 [H, W, 3] BGR -> fp32 [H, W, 3] BGR."""
    lib = _lib.load()
    h, w = mask_blur.shape
    assert image.shape == (3, h, w) and pred.shape == (h, w, 3) and image.is_contiguous() and pred.is_contiguous()
    out = torch.empty((h, w, 3), device=mask_blur.device, dtype=torch.float32)
    _lib.check(lib.mobi_blend_frame(_ptr(mask_blur), _ptr(image), _ptr(pred), _ptr(out), h, w, _stream()), "mobi_blend_frame")
    return out
