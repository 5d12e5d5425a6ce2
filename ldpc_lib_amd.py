"""Import alias: the package directory is named `ldpc-lib_amd/` (a hyphen is not importable), so
`import ldpc_lib_amd` loads that directory as the package `ldpc_lib_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ldpc-lib_amd")
_spec = importlib.util.spec_from_file_location("ldpc_lib_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ldpc_lib_amd"] = _mod
_spec.loader.exec_module(_mod)
