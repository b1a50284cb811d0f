"""Import shim: the package directory is named ``vit-tf_amd`` (not a Python identifier), so
``import vit_tf_amd`` resolves here and swaps itself for that package."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'vit-tf_amd')
_spec = importlib.util.spec_from_file_location('vit_tf_amd', os.path.join(_pkg_dir, '__init__.py'),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['vit_tf_amd'] = _mod
_spec.loader.exec_module(_mod)
