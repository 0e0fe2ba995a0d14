"""Host utilities with the reference's names (ldm/util.py): the config plugin mechanism
(`instantiate_from_config`, :76-91), `cat_interleave` (:213-221), `make_contiguous`
(:203-210), plus a small OmegaConf-free YAML loader with `${}` interpolation."""
import importlib
import re

import torch


def get_obj_from_str(string, reload=False):
    """`target:` strings written for the reference (`ldm.…`) resolve to this package."""
    module, cls = string.rsplit(".", 1)
    if module == "ldm" or module.startswith("ldm."):
        module = "mobi_amd." + module
    mod = importlib.import_module(module)
    if reload:
        importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**(config.get("params") or dict()))


def cat_interleave(tensors):
    """[a0, b0, a1, b1, ...] along the batch axis (index-only)."""
    if len(tensors) == 0:
        return tensors
    return torch.stack(list(tensors), dim=1).reshape(-1, *tensors[0].shape[1:])


def make_contiguous(x):
    if isinstance(x, dict):
        return {k: make_contiguous(v) for k, v in x.items()}
    if x is None or isinstance(x, list):
        return x
    return x.to(memory_format=torch.contiguous_format).float()


def exists(x):
    return x is not None


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) else d


_INTERP = re.compile(r"\$\{([^}]+)\}")


def load_config(path, overrides=None):
    """YAML + `${key.sub}` interpolation + dot-list overrides (`a.b=1`), the subset of
    OmegaConf the reference's configs and harness use (inference_test_bench.py:339-341)."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    for item in overrides or []:
        key, val = item.split("=", 1)
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)

    def lookup(key):
        node = cfg
        for p in key.split("."):
            node = node[p]
        return resolve(node)

    def resolve(node):
        if isinstance(node, dict):
            return {k: resolve(v) for k, v in node.items()}
        if isinstance(node, list):
            return [resolve(v) for v in node]
        if isinstance(node, str):
            m = _INTERP.fullmatch(node)
            if m:
                return lookup(m.group(1))
            return _INTERP.sub(lambda mm: str(lookup(mm.group(1))), node)
        return node

    return resolve(cfg)
