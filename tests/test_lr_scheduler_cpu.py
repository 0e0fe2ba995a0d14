"""CPU: the learning-rate schedules `scheduler_config` names (ldm/lr_scheduler.py) against the REFERENCE's own objects
(tests/golden/lr_schedules.npz, made by tests/golden/make_golden_lr.py from /root/reference), bit for bit; the LambdaLR wrapper
and the Lightning-form return value of `configure_optimizers` (ddpm.py:1651-1668)."""
import os
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_lr", os.path.join(HERE, "golden", "make_golden_lr.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CASES


def test_schedules_equal_the_reference_bit_for_bit():
    from ldm import lr_scheduler as L                     # the reference's spelling (ldm aliases mobi_amd.ldm)
    gold = np.load(os.path.join(HERE, "golden", "lr_schedules.npz"))
    for name, (cls, kw) in _cases().items():
        s = getattr(L, cls)(**kw)
        got = np.asarray([float(s(int(n))) for n in gold[name + "_steps"]], dtype=np.float64)
        assert np.array_equal(got, gold[name]), (name, got, gold[name])
        assert float(getattr(s, "last_f", getattr(s, "last_lr", None))) == got[-1]
    # MObI's schedule in words: 200 linear warm-up steps from 1e-6, then a constant rate
    s = L.LambdaLinearScheduler(warm_up_steps=[200], cycle_lengths=[10000000000000], f_start=[1.e-6], f_max=[1.], f_min=[1.])
    assert s(0) == 1e-6 and s(200) == 1.0 and s(10 ** 9) == 1.0 and 0.49 < s(100) < 0.51
    two = L.LambdaLinearScheduler(warm_up_steps=[10, 5], cycle_lengths=[100, 50], f_start=[0.0, 0.1], f_max=[1.0, 0.5], f_min=[0.2, 0.05])
    assert two.find_in_interval(100) == 0 and two.find_in_interval(101) == 1          # a boundary step belongs to the earlier cycle
    with pytest.raises(IndexError):
        two(151)                                                                      # beyond the last cycle: an error, as there


def test_lambda_lr_and_configure_optimizers_form():
    from mobi_amd import train
    from ldm.models.diffusion.ddpm import LatentDiffusion

    class Opt:
        lr = 2.0
    sch = train.LambdaLR(Opt, lambda n: 0.5 if n < 2 else 1.0)
    assert Opt.lr == 1.0 and sch.get_last_lr() == [1.0]                               # step 0 applied at construction
    sch.step()
    assert Opt.lr == 1.0
    sch.step()
    assert Opt.lr == 2.0 and sch.last_epoch == 2 and sch.base_lrs == [2.0]

    net = torch.nn.Module()
    net.cond_adapter_norm = torch.nn.LayerNorm(4)
    net.cross_modal_attn = torch.nn.Linear(4, 4)
    net.frozen = torch.nn.Linear(4, 4)
    stub = types.SimpleNamespace(model=types.SimpleNamespace(diffusion_model=net), cond_stage_trainable=False, learning_rate=1e-5,
                                 use_scheduler=True,
                                 scheduler_config={"target": "ldm.lr_scheduler.LambdaLinearScheduler",
                                                   "params": {"warm_up_steps": [200], "cycle_lengths": [10000000000000],
                                                              "f_start": [1.e-6], "f_max": [1.], "f_min": [1.]}})
    opts, scheds = LatentDiffusion.configure_optimizers(stub)
    assert len(opts) == 1 and len(scheds) == 1 and scheds[0]["interval"] == "step" and scheds[0]["frequency"] == 1
    opt, sch = opts[0], scheds[0]["scheduler"]
    assert sorted(opt.params) == ["model.diffusion_model.cond_adapter_norm.bias", "model.diffusion_model.cond_adapter_norm.weight",
                                  "model.diffusion_model.cross_modal_attn.bias", "model.diffusion_model.cross_modal_attn.weight"]
    assert opt.lr == 1e-5 * 1e-6                                                      # the warm-up's first factor
    for _ in range(200):
        sch.step()
    assert opt.lr == 1e-5
    stub.use_scheduler = False
    assert isinstance(LatentDiffusion.configure_optimizers(stub), train.AdamW)
