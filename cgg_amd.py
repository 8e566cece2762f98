"""Import alias: `import cgg_amd` == the package in ./betrayed-by-captions_amd (hyphenated name)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = 'betrayed-by-captions_amd'
_pkg = importlib.import_module(_REAL)


class _AliasFinder:
    """Resolve `cgg_amd.x.y` to the already-imported `betrayed-by-captions_amd.x.y` module object."""

    @staticmethod
    def find_spec(name, path=None, target=None):
        if not name.startswith('cgg_amd.'):
            return None
        real = _REAL + name[len('cgg_amd'):]
        mod = importlib.import_module(real)
        sys.modules[name] = mod
        return importlib.util.spec_from_loader(name, loader=_AliasLoader(mod))


class _AliasLoader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


import importlib.util  # noqa: E402

sys.meta_path.insert(0, _AliasFinder)
sys.modules[__name__] = _pkg
