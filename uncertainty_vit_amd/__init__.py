"""Importable alias for the package directory `uncertainty-vit_amd/` (a hyphen cannot be
imported): `import uncertainty_vit_amd.modeling_cyclical` resolves inside that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "uncertainty-vit_amd")]
from .native import lib, build  # noqa: E402,F401
