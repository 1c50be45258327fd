"""Import alias: `import vognet_amd` -> the package directory `vognet-pytorch_amd/`
(a hyphen is not importable by name)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("vognet-pytorch_amd")
sys.modules[__name__] = _pkg
