"""Importable alias of the `gpu-wah_amd/` package (its directory name has a hyphen)."""
import importlib
import sys

_pkg = importlib.import_module("gpu-wah_amd")
sys.modules[__name__] = _pkg
