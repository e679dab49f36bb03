"""``src`` alias root.

The reference imports its own package under two names: ``acoustic_locating_vq_vae...`` everywhere and
``src.acoustic_locating_vq_vae...`` in vq_vae/modules/residual_stack.py:28, so whole-module pickles record
``Residual`` under the ``src.`` path.  Both names must resolve to the SAME module objects
here (one class identity), so instead of loading the files twice a meta-path finder aliases
``src.acoustic_locating_vq_vae[.x.y]`` to ``acoustic_locating_vq_vae[.x.y]``.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_PREFIX = "src.acoustic_locating_vq_vae"
_REAL = "acoustic_locating_vq_vae"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self._real_name = real_name

    def create_module(self, spec):
        module = importlib.import_module(self._real_name)
        self._real_spec = module.__spec__
        return module

    def exec_module(self, module):  # already executed under its real name; undo importlib's __spec__ rebinding
        module.__spec__ = self._real_spec


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _PREFIX or fullname.startswith(_PREFIX + "."):
            real = _REAL + fullname[len(_PREFIX):]
            real_spec = importlib.util.find_spec(real)
            if real_spec is None:
                return None
            spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=real_spec.submodule_search_locations is not None)
            return spec
        return None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
