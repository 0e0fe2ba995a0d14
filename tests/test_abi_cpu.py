"""CPU: the C-ABI library builds, loads, exports every symbol include/mobi_engine.h declares, the
ctypes struct layouts match, and the product path refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mobi_amd import _lib, build
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    from mobi_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mobi_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mobi_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)


def test_struct_layouts_match(lib):
    from mobi_amd import _lib
    for sid, cls in _lib.STRUCT_IDS.items():
        assert lib.mobi_struct_size(sid) == C.sizeof(cls), cls.__name__
    assert lib.mobi_struct_size(99) == 0
    assert lib.mobi_abi_version() == 6 == _lib.ABI_VERSION
    assert lib.mobi_error_string(-2) == b"unsupported shape or mode"


def test_argument_validation_without_gpu(lib):
    """Validation happens before any launch, so it is checkable on a CPU-only host."""
    from mobi_amd import _lib
    p = _lib.IgemmParams()
    assert lib.mobi_igemm(C.byref(p), None) == -1                      # null pointers
    p.src0 = p.weight = p.out = 4096
    p.c0, p.batch, p.hin, p.win, p.hout, p.wout, p.kh, p.kw, p.stride, p.groups = 64, 4, 8, 8, 8, 8, 3, 3, 1, 1
    p.pad_h = p.pad_w = 1
    p.cout = p.n_packed = 64
    p.scale, p.ln_svec = 1.0, 4096
    assert lib.mobi_igemm(C.byref(p), None) == -2                      # LayerNorm fold: 1 x 1 launches only
    p.ln_svec, p.groups = None, 3
    assert lib.mobi_igemm(C.byref(p), None) == -1                      # groups: a divisor of batch
    assert lib.mobi_igemm_sync_bytes(C.byref(p), 1) == 0 and lib.mobi_igemm_sync_bytes(C.byref(p), 4) == 2 * 4   # 256 rows x 64 channels
    a = _lib.AttentionParams()
    a.q = a.k = a.vt = a.out = 16
    a.images = a.heads = a.tq = a.tk = 1
    a.dh, a.dtype = 12, 0
    assert lib.mobi_attention(C.byref(a), None) == -2                  # dh % 8 != 0
    a.dh = 168
    assert lib.mobi_attention(C.byref(a), None) == -2                  # dh > 160
    g = _lib.GroupNormParams()
    g.src0 = g.out = g.ws = g.gamma = g.beta = 16
    g.c0, g.batch, g.hw = 48, 1, 4
    assert lib.mobi_groupnorm(C.byref(g), None) == -2                  # channels % 32 != 0
    assert lib.mobi_groupnorm_workspace_bytes(2, 4096) == 2 * 64 * 32 * 2 * 4
    assert lib.mobi_groupnorm_workspace_bytes(2, 256) == 2 * 32 * 32 * 2 * 4     # (at least 32 chunks per image: the chunked kernel)
    g.c0, g.src_f32, g.c1, g.src1 = 64, 1, 32, 16
    assert lib.mobi_groupnorm(C.byref(g), None) == -2                  # an fp32 source is one source
    g.c1, g.src1, g.out_mode = 0, None, 4
    assert lib.mobi_groupnorm(C.byref(g), None) == -1                  # out_mode 0 .. 3
    assert lib.mobi_split_f32(16, 16, 4, 64, 4, 0, None) == -1         # parts 2 | 3
    assert lib.mobi_split_f32(16, 16, 4, 60, 2, 0, None) == -1         # channels % 8
    assert lib.mobi_split_f32(16, 24, 4, 64, 2, 0, None) == -4         # 16-byte alignment
    g.src_f32, g.out_mode, g.c0 = 0, 0, 48
    assert lib.mobi_groupnorm_workspace_bytes(0, 10) == 0
    assert lib.mobi_tile_weights(16, 4096, 24, 32, None) == -2         # rows % 16
    assert lib.mobi_tile_weights(16, 4096, 32, 48, None) == -2         # k % 32
    assert lib.mobi_tile_weights(16, 4104, 32, 32, None) == -4         # 16-byte alignment
    assert lib.mobi_tile_weights(None, 4096, 32, 32, None) == -1
    assert lib.mobi_groupnorm_bwd_workspace_floats(2, 4096, 320) == 2 * 32 * 4 + 2 * 16 * 2 * 320
    rc = _lib.RowChainParams()                                          # mobi_row_chain: shape, program and pointer checks
    rc.dtype, rc.channels, rc.images, rc.rows_per_image, rc.nprog = 0, 640, 2, 1024, 1
    assert lib.mobi_row_chain(C.byref(rc), None) == -2                 # C = 320 only
    rc.channels, rc.rows_per_image = 320, 100
    assert lib.mobi_row_chain(C.byref(rc), None) == -2                 # whole 128-row tiles only
    rc.rows_per_image = 1024
    assert lib.mobi_row_chain(C.byref(rc), None) == -1                 # empty program
    rc.nops[0] = 1
    rc.prog[0][0].code, rc.prog[0][0].flags, rc.prog[0][0].p0, rc.prog[0][0].bias = _lib.CH_PRODUCT, _lib.CH_TO_S, 16, 16
    assert lib.mobi_row_chain(C.byref(rc), None) == -2                 # a flag combination the kernel has no epilogue for
    rc.prog[0][0].flags, rc.prog[0][0].p1 = _lib.CH_STORE, 32
    assert lib.mobi_row_chain(C.byref(rc), None) == -1                 # prefetch pointer without a next product
    assert lib.mobi_row_chain_weight_bytes(320) == 200 * 1024 and lib.mobi_row_chain_weight_bytes(640) == 0


def test_no_cpu_fallback():
    """The product modules must fail loudly off-GPU instead of computing on the CPU."""
    from mobi_amd import _lib, ops
    from mobi_amd.ldm.modules.diffusionmodules.openaimodel import ResBlock
    rb = ResBlock(32, 64, 0.0, out_channels=32)
    with pytest.raises(_lib.EngineUnavailable):
        rb(torch.zeros(1, 32, 4, 4), torch.zeros(1, 64))
    with pytest.raises(_lib.EngineUnavailable):
        ops.groupnorm(torch.zeros(1, 2, 2, 32, dtype=torch.float16), torch.ones(32), torch.zeros(32), 1e-5, True)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mobi_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_state_dict_keys_match_reference_layout():
    """Same parameter names / shapes as the reference modules (checked against the oracle's
    shape tables, themselves pinned to the reference by the n_params golden)."""
    from oracle.unet import UNetConfig, unet_param_shapes
    from oracle.vae import VAEConfig, vae_param_shapes
    from mobi_amd.ldm.models.autoencoder import AutoencoderKL
    from mobi_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    net = UNetModel(image_size=8, in_channels=9, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
                    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                    transformer_depth=1, context_dim=768, legacy=False, bbox_cond=True, use_camera=True,
                    use_lidar=True)
    want = unet_param_shapes(UNetConfig(model_channels=64))
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == {k: tuple(s) for k, s in want.items()}
    for lidar in (False, True):
        dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=2 if lidar else 3,
                  out_ch=2 if lidar else 3, ch=32, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[],
                  lidar_adapter=lidar, dropout=0.0)
        vae = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)
        want = vae_param_shapes(VAEConfig(in_channels=dd["in_channels"], out_ch=dd["out_ch"], ch=32,
                                          lidar_adapter=lidar))
        got = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
        assert got == {k: tuple(s) for k, s in want.items()}


def test_split_k_slabs_are_promised_to_groupnorms_only():
    """ops.Deferred (a split-K launch's unsummed slabs) is handed to the NEXT layer only where that layer opens with a GroupNorm
    over its whole input (ResBlock.in_layers[0], SpatialTransformer.norm: openaimodel.py:255-275, attention.py:306 of the
    reference): the routing predicate on the production UNet's layout, block by block."""
    from mobi_amd.ldm.modules.attention import SpatialTransformer
    from mobi_amd.ldm.modules.diffusionmodules.openaimodel import (Downsample, ResBlock, UNetModel, Upsample,
                                                                   _opens_with_groupnorm)
    from mobi_amd.ldm.modules.diffusionmodules.util import Conv2d
    net = UNetModel(image_size=8, in_channels=9, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
                    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                    transformer_depth=1, context_dim=768, legacy=False, bbox_cond=True, use_camera=True,
                    use_lidar=True)
    blocks = list(net.input_blocks) + [net.middle_block] + list(net.output_blocks)
    firsts = [type(b[0]) for b in blocks]
    assert firsts[0] is Conv2d and not _opens_with_groupnorm(blocks[0])           # input_blocks.0: a bare convolution
    n_down = sum(1 for b in net.input_blocks if isinstance(b[0], Downsample))
    assert n_down == 3
    for b in blocks[1:]:
        assert _opens_with_groupnorm(b) == isinstance(b[0], ResBlock), type(b[0])
        assert isinstance(b[0], (ResBlock, Downsample))
    # inside a block: a ResBlock hands over to a SpatialTransformer, never to an Upsample convolution
    seen = set()
    for b in blocks:
        layers = list(b)
        for a, nxt in zip(layers, layers[1:]):
            seen.add((type(a).__name__, type(nxt).__name__, _opens_with_groupnorm(nxt)))
    assert ("ResBlock", "SpatialTransformer", True) in seen
    assert any(k[1] == "Upsample" and k[2] is False for k in seen)
    assert all(k[2] == (k[1] in ("ResBlock", "SpatialTransformer")) for k in seen)
    assert not _opens_with_groupnorm(Upsample(64, True)) and not _opens_with_groupnorm(torch.nn.Identity())


def test_split_source_entry_points_validate_on_the_host(lib):
    """mobi_igemm_slab_count / mobi_igemm_finish / mobi_groupnorm_takes_split / mobi_groupnorm with a split source refuse bad
    arguments before anything is launched (no GPU here: every call below must return on the host)."""
    from mobi_amd import _lib
    assert lib.mobi_igemm_slab_count(None) == -1 and lib.mobi_igemm_finish(None, None) == -1
    assert lib.mobi_groupnorm_takes_split(1280, 0, 16, 64) == 1           # 8 x 8 level, 16-byte pieces
    assert lib.mobi_groupnorm_takes_split(1280, 1280, 16, 256) == 1       # a concat's first source
    assert lib.mobi_groupnorm_takes_split(320, 0, 16, 4096) == 0          # too large for the register form's slab path
    assert lib.mobi_groupnorm_takes_split(320, 0, 8, 1024) == 0           # 8-byte pieces: only with MOBI_GN_SPLIT_PW4=1
    assert lib.mobi_groupnorm_takes_split(48, 0, 1, 64) == 0 and lib.mobi_groupnorm_takes_split(64, 0, 0, 64) == 0
    p = _lib.IgemmParams()
    p.src0 = p.weight = p.out = 16
    p.c0, p.batch, p.hin, p.win, p.hout, p.wout = 640, 2, 8, 8, 8, 8
    p.kh = p.kw = 3
    p.stride, p.pad_h, p.pad_w, p.groups, p.n_packed, p.cout, p.scale, p.dtype = 1, 1, 1, 1, 640, 640, 1.0, 0
    assert lib.mobi_igemm_slab_count(C.byref(p)) == 1                     # no split asked
    p.defer_finish = 1
    assert lib.mobi_igemm_slab_count(C.byref(p)) == -2                    # nothing to defer without a split
    p.split_k, p.ws = 4, 32
    assert lib.mobi_igemm_slab_count(C.byref(p)) == 4
    p.split_k = 64                                                        # 90 k-tiles in 64 ranges of 2: 45 slabs are written
    assert lib.mobi_igemm_slab_count(C.byref(p)) == 45
    p.split_k, p.out_mode = 4, 2
    assert lib.mobi_igemm_slab_count(C.byref(p)) == -2                    # fp32 rows are not what a consumer reconstructs
    p.out_mode, p.defer_finish = 0, 2
    assert lib.mobi_igemm_slab_count(C.byref(p)) == -1
    ss = _lib.SplitSource()
    ss.slabs, ss.count, ss.row_stride = 48, 4, 640
    g = _lib.GroupNormParams()
    g.c0, g.batch, g.hw, g.gamma, g.beta, g.eps, g.out, g.ws, g.dtype = 640, 2, 64, 16, 16, 1e-5, 16, 16, 0
    g.src0_split = C.pointer(ss)
    ss.count = 65
    assert lib.mobi_groupnorm(C.byref(g), None) == -1
    ss.count, ss.row_stride = 4, 642
    assert lib.mobi_groupnorm(C.byref(g), None) == -1                     # rows of slabs are 16-byte aligned
    ss.row_stride, ss.slabs = 640, 52
    assert lib.mobi_groupnorm(C.byref(g), None) == -4
    ss.slabs, g.src_f32 = 48, 1
    assert lib.mobi_groupnorm(C.byref(g), None) == -2
    g.src_f32 = 0
    g.src0_split = None
    assert lib.mobi_groupnorm(C.byref(g), None) == -1                     # neither a tensor nor its slabs


def test_integration_md_stub_matches_the_library(monkeypatch):
    """The ctypes stub INTEGRATION.md shows a maintainer is executed as written (from the repo root): its own ABI-version and
    struct-size assertions run against the built library, so the document cannot fall behind the header again."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "class GroupNormParams" in b]
    assert len(stub) == 1 and "mobi_struct_size(1)" in stub[0] and "mobi_abi_version()" in stub[0]
    monkeypatch.chdir(root)
    ns = {}
    exec(compile(stub[0], "INTEGRATION.md", "exec"), ns)                  # defines the struct, runs the layout assertions
    assert callable(ns["groupnorm_silu"])


def test_config_loader_and_instantiate(tmp_path):
    from mobi_amd.ldm.util import instantiate_from_config, load_config
    y = tmp_path / "c.yaml"
    y.write_text("latent_size: 8\nuse_lidar: true\nmodel:\n  target: ldm.modules.diffusionmodules.openaimodel.UNetModel\n"
                 "  params:\n    image_size: ${latent_size}\n    in_channels: 9\n    out_channels: 4\n"
                 "    model_channels: 32\n    attention_resolutions: [4, 2, 1]\n    num_res_blocks: 1\n"
                 "    channel_mult: [1, 2]\n    num_heads: 4\n    use_spatial_transformer: true\n"
                 "    context_dim: 768\n    legacy: false\n    bbox_cond: true\n    use_lidar: ${use_lidar}\n")
    cfg = load_config(str(y), ["model.params.num_res_blocks=2"])
    assert cfg["model"]["params"]["image_size"] == 8 and cfg["model"]["params"]["use_lidar"] is True
    net = instantiate_from_config(cfg["model"])
    assert type(net).__module__ == "mobi_amd.ldm.modules.diffusionmodules.openaimodel" and net.multimodal


def test_schedules_bit_exact_host_side():
    """The product's own host-side schedule code against the reference golden tables."""
    import numpy as np
    from tests.golden_cases import load
    from mobi_amd.ldm.modules.diffusionmodules import util as U
    g = load("schedule_tables")
    assert np.array_equal(U.make_beta_schedule("linear", 1000, 0.00085, 0.012), g["betas_f64"].numpy())
    for S in (10, 50, 250, 30):
        assert np.array_equal(U.make_ddim_timesteps("uniform", S, 1000, verbose=False),
                              g[f"ddim_timesteps_S{S}"].numpy())
    ac = g["ddpm_alphas_cumprod"]
    for S in (10, 50):
        for eta in (0.0, 1.0):
            ts = U.make_ddim_timesteps("uniform", S, 1000, verbose=False)
            sig, a, ap = U.make_ddim_sampling_parameters(ac, ts, eta, verbose=False)
            tag = f"S{S}_eta{int(eta)}"
            assert np.array_equal(sig, g[f"ddim_sigmas_{tag}"].numpy())
            assert np.array_equal(a, g[f"ddim_alphas_{tag}"].numpy())
            assert np.array_equal(ap, g[f"ddim_alphas_prev_{tag}"].numpy())


def test_every_product_module_imports():
    import importlib
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mobi_amd")):
        for f in files:
            if f.endswith(".py") and f != "build.py":
                rel = os.path.relpath(os.path.join(dirpath, f), ROOT)[:-3].replace(os.sep, ".")
                if rel.endswith(".__init__"):
                    rel = rel[:-9]
                importlib.import_module(rel)


CONFIGS = ["mobi_nusc-mini_256", "mobi_nusc-mini_512", "mobi_nusc_256", "mobi_nusc_512", "mobi_nusc_all-classes_256",
           "mobi_nusc_all-classes_512", "range_autoencoder", "pbe"]


@pytest.mark.skipif(not os.path.exists("/root/reference/configs/mobi_nusc_512.yaml"), reason="reference tree not present")
@pytest.mark.parametrize("name", CONFIGS)
def test_restated_configs_equal_the_reference(name):
    """configs/*.yaml (the files the GPU box can name) hold exactly the values of the reference's YAML files, before
    and after `${}` resolution, and the reference's own files load through the engine's loader."""
    import yaml
    from mobi_amd.ldm.util import load_config
    with open(f"/root/reference/configs/{name}.yaml") as f:
        ref_raw = yaml.safe_load(f)
    with open(os.path.join(ROOT, "configs", f"{name}.yaml")) as f:
        mine_raw = yaml.safe_load(f)
    assert mine_raw == ref_raw
    assert load_config(os.path.join(ROOT, "configs", f"{name}.yaml")) == load_config(f"/root/reference/configs/{name}.yaml")


@pytest.mark.parametrize("name", [c for c in CONFIGS if c.startswith("mobi_")])
def test_mobi_configs_resolve_onto_engine_classes(name):
    from mobi_amd.ldm.util import get_obj_from_str, load_config
    cfg = load_config(os.path.join(ROOT, "configs", f"{name}.yaml"), ["use_lidar=True"])
    mp_ = cfg["model"]["params"]
    assert get_obj_from_str(cfg["model"]["target"]).__module__ == "mobi_amd.ldm.models.diffusion.ddpm"
    assert get_obj_from_str(mp_["unet_config"]["target"]).__module__ == "mobi_amd.ldm.modules.diffusionmodules.openaimodel"
    for k in ("first_stage_config", "lidar_stage_config"):
        assert get_obj_from_str(mp_[k]["target"]).__module__ == "mobi_amd.ldm.models.autoencoder"
    assert get_obj_from_str(mp_["cond_stage_config"]["target"]).__module__ == "mobi_amd.ldm.modules.encoders.modules"
    assert mp_["image_size"] == cfg["latent_size"] == cfg["image_height"] // 8
    assert mp_["unet_config"]["params"]["model_channels"] == 320 and mp_["cond_stage_key"] == ["ref_image", "ref_bbox"]


def test_shipped_library_has_no_scratch_and_no_dev_kernels():
    """Every kernel of the shipped library keeps its state in registers (no scratch_load / scratch_store in the gfx950
    code objects: a spilling main loop was the round-1 library's slowest igemm path) and the A/B partner kernels of the
    development build (-DMOBI_DEV) are not in it."""
    import shutil
    import subprocess
    import tempfile
    from mobi_amd import _lib, build
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(bundler) and os.path.exists(objdump)):
        pytest.skip("ROCm LLVM tools not present")
    assert _lib.load().mobi_build_info() == 0
    objs = [os.path.join(build.OBJ, f) for f in os.listdir(build.OBJ) if f.endswith(".o")]
    assert len(objs) >= 7
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, os.path.basename(o) + ".co")
            if subprocess.run([objcopy, f"--dump-section=.hip_fatbin={fat}", o], capture_output=True).returncode != 0:
                continue                                         # a host-only object (tuning.o has no kernels)
            r = subprocess.run([bundler, "--unbundle", "--type=o", f"--input={fat}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            assert r.returncode == 0 and os.path.getsize(co) > 0, r.stderr
            asm = subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout
            assert "s_endpgm" in asm
            assert "scratch_load" not in asm and "scratch_store" not in asm, o
            for dev_only in ("igemm_glds_kernel", "attention_sp_kernel"):
                assert dev_only not in asm, (o, dev_only)
            if o.endswith(("igemm.o", "attention.o")):
                assert "v_mfma_f32_16x16x32" in asm or "v_mfma_f32_32x32x16" in asm


def test_igemm_plan_routes_small_problems_at_the_boundary(lib, monkeypatch):
    """The launch plan is host logic (no launch): `mobi_igemm_kernel_variant` / `mobi_igemm_plan_splits` at the rows either
    side of the small-problem kernel's caps (1.8 GFLOP of 2 M N K, 1.0 GFLOP at K = 320), its shape rules, and the A/B knobs."""
    from mobi_amd import _lib
    SMALL, RING128 = 7, 4

    def plan(rows, k, n, **kw):
        p = _lib.IgemmParams()
        p.src0 = p.weight = p.out = 4096
        p.c0, p.batch, p.hin, p.win, p.hout, p.wout = k, 1, rows, 1, rows, 1
        p.kh = p.kw = p.stride = p.groups = 1
        p.n_packed = p.cout = n
        p.scale, p.dtype = 1.0, 1
        for key, v in kw.items():
            setattr(p, key, v)
        return lib.mobi_igemm_kernel_variant(C.byref(p)), lib.mobi_igemm_plan_splits(C.byref(p))

    for key in ("MOBI_IGEMM_SMALL", "MOBI_IGEMM_SMALL_MFLOP", "MOBI_IGEMM_SMALL_CONV_M", "MOBI_IGEMM_WM", "MOBI_IGEMM_WIDE", "MOBI_IGEMM_SM",
                "MOBI_IGEMM_SPLIT_ROUND4"):
        monkeypatch.delenv(key, raising=False)
    lib.mobi_tuning_reload()
    try:
        assert plan(512, 1280, 1280) == (SMALL, 1)                       # 1.68 GFLOP: the 8 x 8 level's linears, no split-K slabs
        assert plan(544, 1280, 1280)[0] == SMALL                         # 1.78 GFLOP, ragged rows
        assert plan(576, 1280, 1280)[0] != SMALL                         # 1.89 GFLOP: back on the LDS-ring kernel (+ its split-K plan)
        assert plan(4096, 320, 320)[0] == SMALL                          # K = 320: 0.84 GFLOP
        assert plan(8192, 320, 320)[0] == RING128                        # K = 320: 1.68 GFLOP is above that width's cap
        assert plan(256, 1280, 1280, c1=1280, src1=4096)[0] == SMALL     # two sources, whole 80-channel batches in each
        assert plan(256, 96, 1280, c1=224, src1=4096)[0] != SMALL        # a batch would straddle the sources (c0 % 80)
        assert plan(128, 1280, 1280, kh=3, kw=3, pad_h=1, pad_w=1)[0] != SMALL       # 3 x 3: measured slower, not routed ...
        monkeypatch.setenv("MOBI_IGEMM_SMALL_CONV_M", "256")
        lib.mobi_tuning_reload()
        assert plan(128, 1280, 1280, kh=3, kw=3, pad_h=1, pad_w=1)[0] == SMALL       # ... unless asked for (A/B)
        assert plan(128, 1280, 1280, kh=3, kw=3, pad_h=0, pad_w=0)[0] != SMALL
        monkeypatch.delenv("MOBI_IGEMM_SMALL_CONV_M")
        lib.mobi_tuning_reload()
        assert plan(256, 1296 - 16, 1280)[0] == SMALL and plan(256, 1024, 1280)[0] != SMALL     # K % 320
        assert plan(256, 1280, 1296)[0] != SMALL                         # N % 32
        assert plan(256, 1280, 1280, epilogue=1, n_packed=2560)[0] != SMALL                       # GEGLU
        assert plan(256, 1280, 1280, out_mode=1)[0] != SMALL             # transposed output
        assert plan(256, 1280, 1280, out_mode=2)[0] == SMALL             # fp32 rows
        monkeypatch.setenv("MOBI_IGEMM_SMALL", "0")
        lib.mobi_tuning_reload()
        assert plan(256, 1280, 1280) == (RING128, 1)                      # 20 k-tiles: never worth a second launch (round 5)
        assert plan(256, 2560, 1280) == (RING128, 4)                      # 40 k-tiles on 16 tiles: 5 by the arithmetic, 4 by the reduce's rounds
        assert plan(1024, 5120, 1280)[1] == 4 and plan(512, 5120, 1280)[1] == 8     # 64 tiles: 4; 32 tiles still want 8
        assert plan(4096, 2560, 1280)[1] == 1 and plan(4096, 5120, 1280)[1] == 2     # one full wave of blocks: 40 k-tiles unsplit, 80 in two
        assert plan(1024, 1280, 1280, kh=3, kw=3, pad_h=1, pad_w=1, hin=32, win=32, hout=32, wout=32)[1] == 8   # 180 k-tiles: 8 stay 8
        monkeypatch.setenv("MOBI_IGEMM_SPLIT_ROUND4", "0")
        lib.mobi_tuning_reload()
        assert plan(256, 1280, 1280)[1] == 2 and plan(256, 2560, 1280)[1] == 5      # round 4's plan (A/B)
        monkeypatch.delenv("MOBI_IGEMM_SPLIT_ROUND4")
        lib.mobi_tuning_reload()
        monkeypatch.setenv("MOBI_IGEMM_SMALL", "32")
        lib.mobi_tuning_reload()
        assert plan(65536, 320, 320)[0] == SMALL                         # forced: whatever the size
        monkeypatch.delenv("MOBI_IGEMM_SMALL")
        monkeypatch.setenv("MOBI_IGEMM_WM", "2")                         # a forced tile geometry keeps the launch on the LDS kernels
        lib.mobi_tuning_reload()
        assert plan(256, 1280, 1280)[0] == RING128
    finally:
        monkeypatch.undo()
        lib.mobi_tuning_reload()
