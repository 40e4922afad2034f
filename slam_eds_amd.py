"""Importable alias of the hyphenated package directory ``slam-eds_amd/``."""
import importlib as _il
import sys as _sys

_pkg = _il.import_module("slam-eds_amd")
_sys.modules[__name__] = _pkg
