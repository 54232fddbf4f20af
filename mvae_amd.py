"""Import alias: ``import mvae_amd`` -> the package in ``multimodal-vae-public_amd/`` (a directory
name with a hyphen cannot appear in an ``import`` statement).

``mvae_amd`` and every ``mvae_amd.x.y`` resolve to the SAME module objects as
``multimodal-vae-public_amd.x.y``: a meta-path finder maps the alias names onto the real ones.  (Aliasing only
the top-level name let ``from mvae_amd.engine import ...`` execute engine.py a second time under the alias
name -- two copies of every class, and ``isinstance`` checks that depend on import order.)"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_ALIAS = __name__
_REAL = 'multimodal-vae-public_amd'
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + '.'):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])     # the one real module object

    def exec_module(self, module):
        pass                                                                 # already executed under its real name


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name == _REAL or _name.startswith(_REAL + '.'):
        sys.modules[_ALIAS + _name[len(_REAL):]] = _mod
sys.modules[_ALIAS] = _pkg
