"""GPU: a denoising step captured in a HIP graph (mobi_amd/graph.py) replays the very launches of the eager path:
its results must be BIT-IDENTICAL, across steps, classifier-free guidance, stochastic (eta = 1) steps, PLMS, a change
of the conditioning tokens between runs (refreshed in place, no re-capture) and a change of the weights (re-capture)."""
import pytest
import torch

from oracle import sampler as osampler, unet as ounet, weights as W
from tests.test_gpu_models import _unet

pytestmark = pytest.mark.gpu


def _setup(dtype=torch.float16, mc=64, side=16, b=4):
    import mobi_amd
    mobi_amd.set_engine_dtype(dtype)
    cfg = ounet.UNetConfig(model_channels=mc)
    net = _unet(cfg, side)
    net.load_state_dict(W.synth_state_dict(ounet.unet_param_shapes(cfg), 9))
    net = net.cuda()
    sch = osampler.Schedule(10)

    class Model(torch.nn.Module):                     # an nn.Module: the graph can find the transformer blocks
        num_timesteps = 1000

        def __init__(self):
            super().__init__()
            self.net = net
            self.register_buffer("betas", torch.from_numpy(sch.buffers["betas"]))
            self.register_buffer("alphas_cumprod", torch.from_numpy(sch.buffers["alphas_cumprod"]))
            self.register_buffer("alphas_cumprod_prev", torch.from_numpy(sch.buffers["alphas_cumprod_prev"]))

        @property
        def device(self):
            return self.betas.device

        def apply_model(self, x, t, c):
            return self.net(x, t, context=c)

    inputs = dict(x_T=W.synth_input("g.x_T", (b, 4, side, side)).cuda(), inp=W.synth_input("g.inp", (b, 4, side, side)).cuda(),
                  msk=(W.synth_input("g.mask", (b, 1, side, side)) > 0).float().cuda(),
                  cond=W.synth_input("g.cond", (b, 2, 768)).cuda(), uc=W.synth_input("g.uc", (b, 2, 768)).cuda(),
                  cond2=W.synth_input("g.cond2", (b, 2, 768)).cuda())
    return Model().cuda(), inputs, (b, side)


def _ddim(model, i, shape, graph, scale=1.0, eta=0.0, cond="cond", S=10, noise=None):
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    b, side = shape
    s = DDIMSampler(model, graph=graph) if not isinstance(graph, DDIMSampler) else graph
    out, inter = s.sample(S=S, batch_size=b, shape=[4, side, side], conditioning=i[cond], verbose=False, eta=eta,
                          x_T=i["x_T"], unconditional_guidance_scale=scale, unconditional_conditioning=i["uc"],
                          log_every_t=3, step_noise=noise,
                          test_model_kwargs={"inpaint_image": i["inp"], "inpaint_mask": i["msk"]})
    return out, inter, s


@pytest.mark.parametrize("scale", [1.0, 5.0])
def test_ddim_graph_bit_identical(scale):
    model, i, shape = _setup()
    ref, rint, _ = _ddim(model, i, shape, False, scale)
    got, gint, s = _ddim(model, i, shape, True, scale)
    assert len(s._step_graphs) == 1
    assert torch.equal(got, ref)
    assert len(gint["pred_x0"]) == len(rint["pred_x0"])
    for a, b in zip(gint["pred_x0"] + gint["x_inter"], rint["pred_x0"] + rint["x_inter"]):
        assert torch.equal(a, b)                      # intermediates are copies, not the graph's static buffers
    # a second run with OTHER conditioning tokens reuses the captured graph (context terms refreshed in place)
    ref2, _, _ = _ddim(model, i, shape, False, scale, cond="cond2")
    got2, _, s = _ddim(model, i, shape, s, scale, cond="cond2")
    assert len(s._step_graphs) == 1 and torch.equal(got2, ref2) and not torch.equal(got2, got)
    # and back
    got3, _, _ = _ddim(model, i, shape, s, scale)
    assert torch.equal(got3, ref)


def test_graph_context_terms_are_private_to_the_graph():
    """A captured step reads the transformer blocks' loop-invariant terms (reference vector, adapter keys / values) at
    fixed addresses.  Other users of the same model between two replays -- an eager run with OTHER tokens, a second graph
    with guidance (token batch 2N), the PLMS sampler -- must not change what the first graph reads (one slot per token
    buffer in `BasicTransformerBlock._context_terms`; round 2 had one shared slot)."""
    from mobi_amd.ldm.models.diffusion.plms import PLMSSampler
    model, i, shape = _setup()
    b, side = shape
    ref, _, _ = _ddim(model, i, shape, False)
    got, _, s = _ddim(model, i, shape, True)
    assert torch.equal(got, ref)
    # eager run, other tokens, same shapes
    ref_other, _, _ = _ddim(model, i, shape, False, cond="cond2")
    assert torch.equal(_ddim(model, i, shape, s)[0], ref)
    # a second graph on the same sampler: guidance 5 -> token buffer [2N, 2, 768]; then the first one again
    ref5, _, _ = _ddim(model, i, shape, False, 5.0, cond="cond2")
    got5, _, s = _ddim(model, i, shape, s, 5.0, cond="cond2")
    assert torch.equal(got5, ref5) and len(s._step_graphs) == 2
    assert torch.equal(_ddim(model, i, shape, s)[0], ref)
    # another sampler object (PLMS, its own graphs) with other tokens in between
    p = PLMSSampler(model, graph=True)
    p.sample(S=4, batch_size=b, shape=[4, side, side], conditioning=i["cond2"], verbose=False, x_T=i["x_T"],
             unconditional_guidance_scale=5.0, unconditional_conditioning=i["uc"], inpaint_image=i["inp"],
             inpaint_mask=i["msk"])
    assert torch.equal(_ddim(model, i, shape, s)[0], ref)
    assert torch.equal(_ddim(model, i, shape, s, 5.0, cond="cond2")[0], ref5)
    # a partial batch (its own graph and slots), then the full batch again
    half = {k: v[:2] for k, v in i.items()}
    refh, _, _ = _ddim(model, half, (2, side), False)
    assert torch.equal(_ddim(model, half, (2, side), s)[0], refh)
    assert torch.equal(_ddim(model, i, shape, s)[0], ref)
    assert torch.equal(ref_other, _ddim(model, i, shape, s, cond="cond2")[0])


def test_ddim_graph_stochastic_steps():
    model, i, shape = _setup()
    b, side = shape
    noise = W.synth_input("g.sn", (10, b, 4, side, side)).cuda()
    ref, _, _ = _ddim(model, i, shape, False, eta=1.0, noise=noise)
    got, _, _ = _ddim(model, i, shape, True, eta=1.0, noise=noise)
    assert torch.equal(got, ref)


def test_p_sample_ddim_graph_and_weight_change():
    """The public per-step entry point (bench.py's loop) and invalidation when the weights are replaced."""
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    model, i, shape = _setup()
    b, side = shape
    kw = {"test_model_kwargs": {"inpaint_image": i["inp"], "inpaint_mask": i["msk"]}}

    def steps(s):
        s.make_schedule(10, ddim_eta=0.0, verbose=False)
        x, outs = i["x_T"], []
        for k, step in enumerate(reversed(s.ddim_timesteps.tolist())):
            ts = torch.full((b,), step, device="cuda", dtype=torch.long)
            x, p0 = s.p_sample_ddim(x, i["cond"], ts, index=9 - k, **kw)
            outs.append((x, p0))
            if k == 3:
                break
        return outs

    eager, graphed = steps(DDIMSampler(model, graph=False)), steps(DDIMSampler(model, graph=True))
    for (x0, p0), (x1, p1) in zip(eager, graphed):
        assert torch.equal(x0, x1) and torch.equal(p0, p1)
    assert graphed[0][0].data_ptr() != graphed[1][0].data_ptr()           # copies, not the static output buffer
    # new weights: the captured graph reads the OLD packed copies -> must be dropped, results follow the new weights
    s = DDIMSampler(model, graph=True)
    before = steps(s)[-1][0]
    cfg = ounet.UNetConfig(model_channels=64)
    model.net.load_state_dict(W.synth_state_dict(ounet.unet_param_shapes(cfg), 10))
    after_g = steps(s)[-1][0]
    after_e = steps(DDIMSampler(model, graph=False))[-1][0]
    assert torch.equal(after_g, after_e) and not torch.equal(after_g, before)


@pytest.mark.parametrize("scale", [1.0, 5.0])
def test_plms_graph_bit_identical(scale):
    from mobi_amd.ldm.models.diffusion.plms import PLMSSampler
    model, i, shape = _setup()
    b, side = shape

    def run(graph):
        s = PLMSSampler(model, graph=graph)
        return s.sample(S=10, batch_size=b, shape=[4, side, side], conditioning=i["cond"], verbose=False, x_T=i["x_T"],
                        unconditional_guidance_scale=scale, unconditional_conditioning=i["uc"],
                        inpaint_image=i["inp"], inpaint_mask=i["msk"])[0]

    assert torch.equal(run(True), run(False))


def test_q_sample_device_gather():
    """DDPM.q_sample (ddpm.py:284-287 of the reference) with a ragged per-image t, bit-exact vs torch's own ops."""
    from mobi_amd import ops
    g = torch.Generator().manual_seed(3)
    x0, nz = torch.randn(5, 4, 8, 8, generator=g), torch.randn(5, 4, 8, 8, generator=g)
    t = torch.tensor([0, 999, 500, 21, 981])
    sch = osampler.Schedule(10)
    sa = torch.from_numpy(sch.buffers["sqrt_alphas_cumprod"])
    s1 = torch.from_numpy(sch.buffers["sqrt_one_minus_alphas_cumprod"])
    ref = sa[t].view(-1, 1, 1, 1) * x0 + s1[t].view(-1, 1, 1, 1) * nz
    got = ops.q_sample(x0.cuda(), nz.cuda(), t.cuda(), sa.cuda(), s1.cuda())
    assert torch.equal(got.cpu(), ref)
