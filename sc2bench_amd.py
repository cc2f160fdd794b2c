"""Import shim: makes the package directory ``sc2-benchmark_amd/`` importable as ``sc2bench_amd``."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sc2-benchmark_amd')
_spec = importlib.util.spec_from_file_location('sc2bench_amd', os.path.join(_pkg_dir, '__init__.py'),
                                               submodule_search_locations=[_pkg_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules['sc2bench_amd'] = _module
_spec.loader.exec_module(_module)
