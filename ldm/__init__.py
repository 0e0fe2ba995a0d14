"""`ldm` -- the import name the reference's configs and harness use (`target: ldm.models.diffusion.ddpm.LatentDiffusion`,
`from ldm.models.diffusion.ddim import DDIMSampler`, scripts/inference_test_bench.py:18-23 of the reference).

This package holds no code: it aliases `mobi_amd.ldm`.  `ldm` and every `ldm.x.y` resolve to the very module OBJECTS
of `mobi_amd.ldm.x.y` (not second copies: one set of classes, one set of caches), through a meta-path finder.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = "mobi_amd.ldm"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "ldm" and not fullname.startswith("ldm."):
            return None
        real = _REAL + fullname[3:]
        try:
            spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if spec is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=spec.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[3:])      # the existing module object

    def exec_module(self, module):
        pass                                                       # already executed under its real name


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_impl = importlib.import_module(_REAL)
sys.modules[__name__] = _impl
