"""Import shim: the package directory is named `zk-franchise-proof-circuit_amd` (not a valid Python identifier);
`import zkcensus_amd` loads it under that alias."""
import importlib.util, os, sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'zk-franchise-proof-circuit_amd')
_spec = importlib.util.spec_from_file_location('zkcensus_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['zkcensus_amd'] = _mod
_spec.loader.exec_module(_mod)
