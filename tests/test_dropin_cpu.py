"""CPU: the reference's import spellings and config flow resolve onto the engine, without a GPU.

  from omegaconf import OmegaConf                       (inference_test_bench.py:7)
  from ldm.util import instantiate_from_config          (:18)
  from ldm.models.diffusion.ddim import DDIMSampler     (:19)
  from ldm.models.diffusion.plms import PLMSSampler     (:20)
  config = OmegaConf.load(...); cli = OmegaConf.from_dotlist(...); config = OmegaConf.merge(config, cli)   (:339-341)
  model = instantiate_from_config(config.model)         (:155)
The call sequence itself (get_input -> sample -> decode_sample -> log_data) runs on the GPU: tests/test_gpu_models.py
::test_harness_flow_reference_spelling."""
import inspect
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ldm_alias_is_the_same_module_objects():
    import ldm                                           # top-level package of the repo root
    import mobi_amd.ldm
    assert ldm is mobi_amd.ldm
    from ldm.util import instantiate_from_config
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.models.autoencoder import AutoencoderKL
    import mobi_amd.ldm.models.diffusion.ddim as real_ddim
    import mobi_amd.ldm.util as real_util
    assert DDIMSampler is real_ddim.DDIMSampler and instantiate_from_config is real_util.instantiate_from_config
    assert sys.modules["ldm.models.diffusion.ddim"] is real_ddim
    assert PLMSSampler.__module__ == "mobi_amd.ldm.models.diffusion.plms"
    assert LatentDiffusion.__module__ == "mobi_amd.ldm.models.diffusion.ddpm"
    assert UNetModel.__module__ == "mobi_amd.ldm.modules.diffusionmodules.openaimodel"
    assert AutoencoderKL.__module__ == "mobi_amd.ldm.models.autoencoder"
    with pytest.raises(ImportError):
        import ldm.no_such_module                        # noqa: F401


def test_omegaconf_flow_of_the_harness():
    omegaconf = pytest.importorskip("omegaconf")
    OmegaConf = omegaconf.OmegaConf
    from ldm.util import instantiate_from_config
    config = OmegaConf.load(os.path.join(ROOT, "configs", "mobi_nusc_256.yaml"))
    cli = OmegaConf.from_dotlist(["ref_mode=id-ref", "use_lidar=True", "latent_size=8", "image_height=64",
                                  "model.params.lidar_stage_config.params.ckpt_path=null",
                                  "model.params.unet_config.params.model_channels=32",
                                  "model.params.first_stage_config.params.ddconfig.ch=32",
                                  "model.params.lidar_stage_config.params.ddconfig.ch=32",
                                  "model.params.cond_stage_config=__is_unconditional__"])
    config = OmegaConf.merge(config, cli)
    # interpolation sees the CLI override; attribute and item access both work
    assert config.model.params.image_size == 8 and config["model"]["params"]["unet_config"]["params"]["image_size"] == 8
    assert config.model.params.first_stage_config.params.ddconfig.resolution == 64
    assert list(config.model.params.cond_stage_key) == ["ref_image", "ref_bbox"]
    assert config.data.params.test.params.ref_mode == "id-ref" if "test" in config.data.params else True
    model = instantiate_from_config(config.model)
    assert type(model).__name__ == "LatentDiffusion" and model.image_size == 8 and model.channels == 4
    assert model.use_camera and model.use_lidar and model.first_stage_key == "inpaint"
    assert list(model.cond_stage_key) == ["ref_image", "ref_bbox"]
    assert tuple(model.learnable_vector.shape) == tuple(model.bbox_uncond_vector.shape) == (1, 1, 768)
    assert model.model.diffusion_model.model_channels == 32
    with model.ema_scope():
        pass


def _params(fn):
    return [(n, p.default) for n, p in inspect.signature(fn).parameters.items() if n != "self"]


def test_signatures_match_the_reference_defaults():
    """Names, order and defaults of the entry points the harness calls (ddim.py:57-112, plms.py:57-79,
    ddpm.py:758, :1420, :1471 of the reference)."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion
    E = inspect.Parameter.empty
    want_sample = [("S", E), ("batch_size", E), ("shape", E), ("conditioning", None), ("callback", None),
                   ("normals_sequence", None), ("img_callback", None), ("quantize_x0", False), ("eta", 0.), ("mask", None),
                   ("x0", None), ("temperature", 1.), ("noise_dropout", 0.), ("score_corrector", None),
                   ("corrector_kwargs", None), ("verbose", True), ("x_T", None), ("log_every_t", 100),
                   ("unconditional_guidance_scale", 1.), ("unconditional_conditioning", None), ("kwargs", E)]
    assert _params(DDIMSampler.sample) == want_sample
    assert _params(LatentDiffusion.log_data) == [("batch", E), ("data", E), ("h_camera", E), ("h_lidar", E),
                                                 ("log_metrics", True), ("return_sample", False), ("split", "train")]
    gi = dict(_params(LatentDiffusion.get_input))
    assert list(gi)[:5] == ["batch", "k", "force_c_encode", "bs", "return_vae_rec"]
    assert gi["force_c_encode"] is False and gi["bs"] is None and gi["return_vae_rec"] is False
    assert [n for n, _ in _params(LatentDiffusion.decode_sample)] == ["sample", "z_lidar"]


@pytest.mark.skipif(not os.path.exists("/root/reference/ldm/models/diffusion/ddim.py"), reason="reference tree not present")
def test_signatures_against_the_reference_source():
    """The same check read from the reference's source text (ast only: nothing of the reference is imported)."""
    import ast
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import LatentDiffusion

    def ref_args(path, cls, fn):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ClassDef) and node.name == cls:
                for f in node.body:
                    if isinstance(f, ast.FunctionDef) and f.name == fn:
                        names = [a.arg for a in f.args.args if a.arg != "self"]
                        defaults = [ast.literal_eval(d) for d in f.args.defaults]
                        return names, defaults
        raise KeyError((cls, fn))

    for path, cls, obj, fn in (("/root/reference/ldm/models/diffusion/ddim.py", "DDIMSampler", DDIMSampler, "sample"),
                               ("/root/reference/ldm/models/diffusion/ddpm.py", "LatentDiffusion", LatentDiffusion, "log_data"),
                               ("/root/reference/ldm/models/diffusion/ddpm.py", "LatentDiffusion", LatentDiffusion, "decode_sample")):
        names, defaults = ref_args(path, cls, fn)
        mine = [(n, d) for n, d in _params(getattr(obj, fn)) if n != "kwargs"]
        assert [n for n, _ in mine][:len(names)] == names, (cls, fn)
        mine_defaults = [d for _, d in mine[:len(names)] if d is not inspect.Parameter.empty]
        assert mine_defaults == defaults, (cls, fn)
