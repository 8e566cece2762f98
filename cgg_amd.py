"""Import alias: `import cgg_amd` == the package in ./betrayed-by-captions_amd (hyphenated name)."""
import importlib
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = 'betrayed-by-captions_amd'
_pkg = importlib.import_module(_REAL)


class _AliasLoader:
    def __init__(self, mod):
        self.mod = mod
        self.spec = mod.__spec__

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        module.__spec__ = self.spec  # keep the real module's own spec


class _AliasFinder:
    """Resolve a lazily imported `cgg_amd.x` to the module object of `betrayed-by-captions_amd.x`."""

    @staticmethod
    def find_spec(name, path=None, target=None):
        if not name.startswith('cgg_amd.'):
            return None
        mod = importlib.import_module(_REAL + name[len('cgg_amd'):])
        return importlib.util.spec_from_loader(name, loader=_AliasLoader(mod))


sys.meta_path.insert(0, _AliasFinder)
for _k in list(sys.modules):
    if _k.startswith(_REAL + '.'):
        sys.modules['cgg_amd' + _k[len(_REAL):]] = sys.modules[_k]
sys.modules[__name__] = _pkg
