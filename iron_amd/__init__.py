"""iron_amd -- MI355X (gfx950) implementation of IRON's stage-2 forward render path.

Module names mirror the reference's `models/` package for this path:
    iron_amd.raytracer      <- models/raytracer.py
    iron_amd.renderer_ggx   <- models/renderer_ggx.py
    iron_amd.rendering_func <- models/rendering_func.py
    iron_amd.fields         <- models/fields.py
    iron_amd.embedder       <- models/embedder.py
`install_as_models()` registers them under those names for `render_surface.py`-style callers.
"""
from __future__ import annotations

import sys
import types

__version__ = "0.1.0"


def install_as_models() -> None:
    """Make `from models.raytracer import ...` (and friends) resolve to this package."""
    import importlib

    pkg = sys.modules.get("models")
    if pkg is None:
        pkg = types.ModuleType("models")
        pkg.__path__ = []  # mark as package
        sys.modules["models"] = pkg
    for name in ("raytracer", "renderer_ggx", "rendering_func", "fields", "embedder", "renderer", "network_conf"):
        mod = importlib.import_module("iron_amd." + name)
        sys.modules["models." + name] = mod
        setattr(pkg, name, mod)
