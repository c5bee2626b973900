"""Import alias.  The package directory is named `dmx-compressor_amd` (the repo layout contract), which is
not a Python identifier; `import dmx_compressor_amd` finds this file, which loads that directory as the
package `dmx_compressor_amd` and replaces itself in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dmx-compressor_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
