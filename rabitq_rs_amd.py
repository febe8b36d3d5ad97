"""Import shim: the package directory is `rabitq-rs_amd/` (hyphenated, as the project
layout prescribes), which Python cannot import by name. This module loads it under the
importable name `rabitq_rs_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rabitq-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "rabitq_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rabitq_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
