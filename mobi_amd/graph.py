"""One denoising step as ONE HIP graph launch.

A UNet forward is ~590 engine launches (ctypes call + parameter struct each); at the 32x32 latent of
`mobi_nusc_256` the GPU finishes them faster than the host can issue them.  `StepGraph` captures the launch
sequence of one step -- UNet forward (both halves with classifier-free guidance) and, for DDIM, the fp32 latent
update -- into a HIP graph once per (shapes, storage type, guidance, sampler-kind) key and replays it per step.

What changes from step to step lives in device buffers the graph reads:
  x        the latent state                    (copied in before a replay)
  ts       int64 timestep vector               (filled)
  coef     {a_t, a_prev, sigma_t, sqrt(1-a_t)} (copied from the run's device table; `mobi_ddim_step.coef_dev`)
  noise    the step's Gaussian draw            (eta > 0 only)
What changes from sampling run to sampling run (conditioning tokens, inpaint latents, mask) is copied into static
buffers when the caller's tensors change (identity / version counter); the transformer blocks then refresh their
loop-invariant context terms IN PLACE (`BasicTransformerBlock._context_terms` keeps persistent buffers), so the
captured pointers stay valid and nothing is re-captured.

The replayed launches are the same kernels with the same arguments as the eager path: results are bit-identical
(tests/test_gpu_graph.py).  PyTorch is used for what it is here for: the capture stream, the private memory pool of
the capture (`torch.cuda.graph`) and `hipGraphLaunch` on the current stream.
"""
import gc
import os
import weakref

import torch
import torch.nn as nn

from . import engine_dtype, ops

ENABLED = os.environ.get("MOBI_STEP_GRAPH", "1") != "0"


def set_enabled(flag):
    global ENABLED
    ENABLED = bool(flag)


def _sig(t):
    return None if t is None else (tuple(t.shape), t.dtype)


def _ident(t):
    return None if t is None else (id(t), t._version, t.data_ptr())


def weights_epoch():
    from .ldm.modules.diffusionmodules.util import WEIGHTS_EPOCH
    return WEIGHTS_EPOCH[0]


def weights_fingerprint(model):
    """Sum of the parameters' version counters (in-place edits such as `p.normal_()` move it); computed once per
    sampling run, not per step."""
    if not isinstance(model, nn.Module):
        return 0
    return sum(p._version for p in model.parameters()) + weights_epoch()


def _context_blocks(model):
    if not isinstance(model, nn.Module):
        return None
    from .ldm.modules.attention import BasicTransformerBlock
    return [m for m in model.modules() if isinstance(m, BasicTransformerBlock)]


class StepGraph:
    """kind = "ddim": eps + latent update -> (x_prev, pred_x0); kind = "eps": UNet forward(s) only ->
    (e_cond, e_uncond | None) (PLMS mixes the eps history eagerly)."""

    def __init__(self, sampler, kind, x, cond, uncond, scale, parts_extra, parts_key, temperature, has_noise):
        # (weak: the sampler owns this object through its graph table -- a strong back-reference would make a cycle whose
        #  collection could destroy a hipGraph in the middle of some later capture)
        self._sampler, self.kind = weakref.ref(sampler), kind
        dev = x.device
        self.cfg = uncond is not None and scale != 1.
        self.scale, self.temperature = float(scale), float(temperature)
        self.parts_key = parts_key
        self.x = torch.empty_like(x)
        self.ts = torch.zeros((x.shape[0],), device=dev, dtype=torch.long)
        self.coef = torch.ones(4, device=dev, dtype=torch.float32)
        self.noise = torch.zeros_like(x) if has_noise else None
        self.extra = [torch.empty(p.shape, device=dev, dtype=torch.float32) for p in parts_extra]
        # conditioning tokens as the UNet sees them: [uncond ; cond] with guidance (ddim.py:180-183 of the reference)
        n = cond.shape[0]
        self.ctx = torch.empty((2 * n if self.cfg else n,) + tuple(cond.shape[1:]), device=dev, dtype=torch.float32)
        self.cond = self.ctx[n:] if self.cfg else self.ctx
        self.uncond = self.ctx[:n] if self.cfg else None
        self._extra_id = self._cond_id = self._refs = None
        self._warm = False
        self.refresh(parts_extra, cond, uncond)
        self._warm = True
        self.x.copy_(x)
        # eager warm-up on a side stream: packs weights, fills every cache the capture must hit
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._body()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # no cyclic garbage collection while the stream is capturing: a collected tensor / graph / event is released
        # through HIP calls that are illegal during capture (torch collects once before the capture starts)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                self.outputs = self._body()
        finally:
            if gc_was_on:
                gc.enable()
        # what the captured launches read of the transformer blocks' loop-invariant terms: this graph's own slot per block
        # (keyed on self.ctx); checked before every run's first replay
        self._term_ptrs = self._context_term_addresses()

    def _context_term_addresses(self):
        blocks = _context_blocks(self.sampler.model)
        return None if blocks is None else [b.context_term_addresses(self.ctx) for b in blocks]

    @property
    def sampler(self):
        s = self._sampler()
        if s is None:
            raise RuntimeError("the sampler that owns this step graph is gone")
        return s

    def _kwargs(self):
        if self.parts_key == "test_model_kwargs":
            return {"test_model_kwargs": {"inpaint_image": self.extra[0], "inpaint_mask": self.extra[1]}}
        return {"rest": self.extra[0]}

    def _body(self):
        s = self.sampler
        e_c, e_u = s._eps(self.x, self.cond, self.ts, self.scale, self.uncond, self._kwargs(),
                          cfg_ctx=self.ctx if self.cfg else None)
        if self.kind == "eps":
            return e_c, e_u
        x_prev, pred, _ = ops.ddim_step(self.x, e_c, e_uncond=e_u, noise=self.noise, cfg_scale=self.scale,
                                        temperature=self.temperature, coef_dev=self.coef)
        return x_prev, pred

    def refresh(self, parts_extra, cond, uncond):
        """Run-level inputs: copied (and the blocks' context terms recomputed in place) only when they changed."""
        eid = tuple(_ident(p) for p in parts_extra)
        if eid != self._extra_id:
            for dst, src in zip(self.extra, parts_extra):
                dst.copy_(src)
            self._extra_id = eid
        cid = (_ident(cond), _ident(uncond) if self.cfg else None)
        if cid != self._cond_id:
            self.cond.copy_(cond)
            if self.cfg:
                self.uncond.copy_(uncond)
            self._cond_id = cid
            if self._warm:
                # loop-invariant context terms: recomputed IN PLACE in this graph's own slot of every block (keyed on
                # self.ctx: other users of the model -- another graph, the PLMS sampler, eager calls -- have slots of their
                # own and cannot change what the captured launches read; weight changes make a new graph, see `get`); the
                # captured addresses must still be the slots'
                blocks = _context_blocks(self.sampler.model)
                if blocks is None:
                    self._body()                 # opaque model: one eager evaluation refreshes every cache
                else:
                    for b in blocks:
                        b._context_terms(self.ctx)
                    if getattr(self, "_term_ptrs", None) is not None and self._context_term_addresses() != self._term_ptrs:
                        raise RuntimeError("a transformer block's context-term buffers moved under a captured step graph")
        self._refs = (parts_extra, cond, uncond)  # keep the callers' tensors alive: their ids / addresses stay unique

    def run(self, x, step, coef_row=None, noise=None):
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.ts.fill_(int(step))
        if coef_row is not None:
            self.coef.copy_(coef_row)
        if self.noise is not None:
            self.noise.copy_(noise)
        self.graph.replay()
        return self.outputs


def usable(x):
    return ENABLED and x.is_cuda and ops._PROFILE is None and not torch.cuda.is_current_stream_capturing()


def get(sampler, kind, x, cond, uncond, scale, kwargs, temperature=1., has_noise=False):
    """The sampler's graph for this call signature (captured on first use), refreshed for the caller's tensors."""
    if "test_model_kwargs" in kwargs:
        kw = kwargs["test_model_kwargs"]
        parts_key, extra = "test_model_kwargs", [kw["inpaint_image"], kw["inpaint_mask"]]
    elif "rest" in kwargs:
        parts_key, extra = "rest", [kwargs["rest"]]
    else:
        raise Exception("kwargs must contain either 'test_model_kwargs' or 'rest' key")
    cfg = uncond is not None and scale != 1.
    key = (kind, _sig(x), _sig(cond), _sig(uncond) if cfg else None, float(scale) if cfg else 1., parts_key,
           tuple(_sig(p) for p in extra), float(temperature), bool(has_noise), engine_dtype(), x.device,
           weights_epoch(), sampler.__dict__.get("_weights_fp", 0))
    cache = sampler.__dict__.setdefault("_step_graphs", {})
    g = cache.get(key)
    if g is None:
        for k in [k for k in cache if k[-2:] != key[-2:]]:      # weights changed: those graphs read freed packs
            del cache[k]
        if len(cache) >= 4:                      # a sampler object rarely sees more than two signatures
            cache.pop(next(iter(cache)))
        g = StepGraph(sampler, kind, x, cond, uncond if cfg else None, scale, extra, parts_key, temperature, has_noise)
        cache[key] = g
    else:
        g.refresh(extra, cond, uncond if cfg else None)
    return g
