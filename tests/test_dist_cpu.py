"""CPU: the N>1 path (object sharding + the one all-gather per batch) with world_size 2 and 3 on gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_objects, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobi_amd import dist as md
    full = {"image": {"GT": torch.arange(n_objects * 3 * 4 * 4, dtype=torch.float32).reshape(n_objects, 3, 4, 4),
                      "meta": "kept"},
            "lidar": {"range_data": -torch.arange(n_objects * 2 * 4 * 4, dtype=torch.float32).reshape(n_objects, 2, 4, 4)}}
    mine = md.shard_batch(full, n_objects)
    lo, hi = md.shard_range(n_objects, rank, world)
    assert mine["image"]["GT"].shape[0] == hi - lo and mine["image"]["meta"] == "kept"
    assert torch.equal(mine["image"]["GT"], full["image"]["GT"][lo:hi])
    # stand-in for sampling + decoding on this rank's objects: any per-object function
    decoded = {"image_sample": mine["image"]["GT"] * 2 + 1, "lidar_sample": mine["lidar"]["range_data"] - 3}
    out = md.gather_decoded(decoded, n_objects)
    ok = torch.equal(out["image_sample"], full["image"]["GT"] * 2 + 1) and \
        torch.equal(out["lidar_sample"], full["lidar"]["range_data"] - 3)
    q.put((rank, bool(ok), tuple(out["image_sample"].shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_objects", [(2, 8), (2, 5), (3, 7)])
def test_shard_and_gather_gloo(world, n_objects):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_objects, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape[0] == n_objects for _, _, shape in res)


def test_shard_ranges_cover_exactly():
    from mobi_amd.dist import shard_range
    for n in (1, 5, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mobi_amd import dist as md
    g = torch.Generator().manual_seed(100 + rank)
    shapes = {"a.to_q.weight": (64, 64), "a.to_out.0.bias": (64,), "b.norm.weight": (64,), "c.connector.weight": (64, 64),
              "d.to_k.weight": (64, 768)}
    mine = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    every = []
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        every.append({k: torch.randn(s, generator=gr) for k, s in shapes.items()})
    want = {k: sum(e[k] for e in every) / world for k in shapes}
    got = md.allreduce_gradients({k: v.clone() for k, v in mine.items()}, bucket_bytes=20000)      # several buckets, one ragged
    ok = all(torch.allclose(got[k], want[k], atol=1e-6) and got[k].shape == want[k].shape for k in shapes)
    tot = md.allreduce_gradients({k: v.clone() for k, v in mine.items()}, average=False)
    ok = ok and all(torch.allclose(tot[k], want[k] * world, atol=1e-5) for k in shapes)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    """The training step's collective (mobi_amd.dist.allreduce_gradients): bucketed, in name order, sum and mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _divergent_worker(rank, world, port, q):
    """Two ranks whose conditioning draws differ (ddpm.py:1049-1052 draws per process): rank 0 took the unconditional
    branch (a gradient for `bbox_uncond_vector`), rank 1 the conditional one (gradients for the box embedder)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    import torch.nn as nn
    from mobi_amd import dist as md
    from mobi_amd.ldm.models.diffusion.ddpm import LatentDiffusion

    class Stand(object):                           # the two methods under test on a stand-in with the attributes they read
        cond_stage_trainable = True
        cond_stage_key = ["ref_image", "ref_bbox"]
        _cond_stage_trainables = LatentDiffusion._cond_stage_trainables
        _complete_cond_stage_grads = LatentDiffusion._complete_cond_stage_grads

    s = Stand()
    emb = nn.Module()
    emb.bbox_proj = nn.Linear(6, 8)
    emb.class_embedder = nn.Linear(3, 8)           # filtered out, as in configure_optimizers
    s.cond_stage_model = types.SimpleNamespace(bbox_embedder=emb)
    s.bbox_uncond_vector = nn.Parameter(torch.zeros(1, 1, 8))
    unet = {"model.diffusion_model.a.cross_modal.weight": torch.full((4, 4), float(rank + 1))}
    ok = True
    # (1) divergent draws: the key sets differ before, agree after; the branch not taken contributed zeros
    named = dict(unet)
    if rank == 0:
        named["bbox_uncond_vector"] = torch.full((1, 1, 8), 2.0)
    else:
        named["cond_stage_model.bbox_embedder.bbox_proj.weight"] = torch.full((8, 6), 4.0)
        named["cond_stage_model.bbox_embedder.bbox_proj.bias"] = torch.full((8,), 6.0)
    md.allreduce_gradients(s._complete_cond_stage_grads(named, torch.device("cpu"), across_ranks=True))
    ok = ok and sorted(named) == ["bbox_uncond_vector", "cond_stage_model.bbox_embedder.bbox_proj.bias",
                                  "cond_stage_model.bbox_embedder.bbox_proj.weight", "model.diffusion_model.a.cross_modal.weight"]
    ok = ok and torch.allclose(named["bbox_uncond_vector"], torch.full((1, 1, 8), 1.0))
    ok = ok and torch.allclose(named["cond_stage_model.bbox_embedder.bbox_proj.weight"], torch.full((8, 6), 2.0))
    ok = ok and torch.allclose(named["cond_stage_model.bbox_embedder.bbox_proj.bias"], torch.full((8,), 3.0))
    ok = ok and torch.allclose(named["model.diffusion_model.a.cross_modal.weight"], torch.full((4, 4), 1.5))
    # (2) both ranks unconditional: the embedder stays absent (DDP leaves `.grad` None, AdamW skips it)
    named = dict(unet)
    named["bbox_uncond_vector"] = torch.full((1, 1, 8), 2.0)
    md.allreduce_gradients(s._complete_cond_stage_grads(named, torch.device("cpu"), across_ranks=True))
    ok = ok and sorted(named) == ["bbox_uncond_vector", "model.diffusion_model.a.cross_modal.weight"]
    # (3) a rank that arrives with another key set is an error on every rank, not a hang / a sum of unrelated tensors
    named = dict(unet)
    if rank == 1:
        named["extra"] = torch.zeros(3)
    try:
        md.allreduce_gradients(named)
        ok = False
    except RuntimeError as e:
        ok = ok and "different gradient sets" in str(e)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_training_step_collective_with_divergent_draws_gloo_world2():
    """ranks whose `u_cond` draws differ used to bring different tensors to the bucketed all-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_divergent_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
