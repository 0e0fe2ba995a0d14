"""`omegaconf` stand-in for images that lack the real package (this one does; SURVEY.md section 5, "Config / flags").

The reference's harness does three things with OmegaConf (scripts/inference_test_bench.py:339-341; main.py:503-505):
`OmegaConf.load(path)`, `OmegaConf.from_dotlist(["a.b=1", ...])`, `OmegaConf.merge(cfg, cli)`, and then reads the result
through attribute / item access with `${a.b}` interpolation resolved lazily (so a CLI override of `latent_size`
reaches every `${latent_size}`).  That subset is implemented here over PyYAML.  If the real omegaconf is installed
and comes first on sys.path it simply wins; this directory is only found when the repo root is on the path.
"""
import copy
import re
from collections.abc import Mapping

import yaml

from .listconfig import ListConfig

__all__ = ["OmegaConf", "DictConfig", "ListConfig"]
_INTERP = re.compile(r"\$\{([^}]+)\}")


class DictConfig(Mapping):
    """Read-mostly view of a nested dict with lazy `${}` resolution against the root."""

    def __init__(self, data=None, root=None):
        object.__setattr__(self, "_data", data if data is not None else {})
        object.__setattr__(self, "_root", root if root is not None else self)

    # -- resolution ----------------------------------------------------------------------
    def _lookup(self, dotted):
        node = self._root._data
        for part in dotted.split("."):
            node = node[part]
        return self._wrap(node)

    def _wrap(self, v):
        if isinstance(v, DictConfig):
            return DictConfig(v._data, self._root)
        if isinstance(v, dict):
            return DictConfig(v, self._root)
        if isinstance(v, (list, tuple)):
            return ListConfig([self._wrap(x) for x in v])
        if isinstance(v, str):
            m = _INTERP.fullmatch(v)
            if m:
                return self._lookup(m.group(1))
            if _INTERP.search(v):
                return _INTERP.sub(lambda mm: str(self._lookup(mm.group(1))), v)
        return v

    # -- mapping protocol ----------------------------------------------------------------
    def __getitem__(self, key):
        return self._wrap(self._data[key])

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return len(self._data)

    def __contains__(self, key):
        return key in self._data

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self._data[key] = _plain(value)

    def __setitem__(self, key, value):
        self._data[key] = _plain(value)

    def get(self, key, default=None):
        return self[key] if key in self._data else default

    def pop(self, key, *default):
        return self._wrap(self._data.pop(key, *default))

    def __repr__(self):
        return f"DictConfig({self._data!r})"


def _plain(v, resolve=False):
    """Plain dict / list containers out of config nodes."""
    if isinstance(v, DictConfig):
        return {k: _plain(v[k] if resolve else v._data[k], resolve) for k in v._data}
    if isinstance(v, dict):
        return {k: _plain(x, resolve) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x, resolve) for x in v]
    return v


def _merge_into(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_into(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


class OmegaConf:
    @staticmethod
    def create(obj=None):
        if isinstance(obj, str):
            obj = yaml.safe_load(obj)
        if isinstance(obj, (list, tuple)):
            return ListConfig(obj)
        return DictConfig(copy.deepcopy(_plain(obj)) if obj is not None else {})

    @staticmethod
    def load(path):
        with open(path) as f:
            return DictConfig(yaml.safe_load(f) or {})

    @staticmethod
    def from_dotlist(dotlist):
        out = {}
        for item in dotlist or []:
            key, _, val = item.partition("=")
            node = out
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = yaml.safe_load(val) if val != "" else None
        return DictConfig(out)

    @staticmethod
    def merge(*configs):
        out = {}
        for c in configs:
            _merge_into(out, _plain(c))
        return DictConfig(out)

    @staticmethod
    def to_container(cfg, resolve=False):
        return _plain(cfg, resolve)

    @staticmethod
    def to_yaml(cfg, resolve=False):
        return yaml.safe_dump(_plain(cfg, resolve), sort_keys=False)

    @staticmethod
    def is_config(obj):
        return isinstance(obj, (DictConfig, ListConfig))
