"""Surface fixture (BUILD CONTAINER ONLY -- imports the reference at /root/reference): inspect.signature of every public
function, class constructor and public method the reference defines in the modules this build mirrors
(models.raytracer, renderer_ggx, rendering_func, fields, embedder, renderer, network_conf).  Data only: names, parameter
names, kinds and defaults (repr) -> tests/golden/surface_signatures.json; tests/test_surface_signatures.py holds
iron_amd.* to it.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_signatures.py
"""
from __future__ import annotations

import importlib
import inspect
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: F401,E402  (registers the placeholder modules and puts the reference on sys.path)

for _n in ("mcubes", "pyhocon", "tinycudann"):
    sys.modules.setdefault(_n, types.ModuleType(_n))

MODULES = ["models.raytracer", "models.renderer_ggx", "models.rendering_func", "models.fields", "models.embedder", "models.renderer",
           "models.network_conf"]


def describe(fn):
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return None
    return [{"name": p.name, "kind": p.kind.name, "default": None if p.default is inspect._empty else repr(p.default)}
            for p in sig.parameters.values()]


def main():
    out = {}
    for modname in MODULES:
        try:
            mod = importlib.import_module(modname)
        except Exception as e:  # network_conf imports fine on CPU; anything that does not is recorded, not emulated
            out[modname] = {"__import_error__": repr(e)}
            continue
        entry = {}
        for name, obj in sorted(vars(mod).items()):
            if name.startswith("_") or getattr(obj, "__module__", None) != modname:
                continue
            if inspect.isfunction(obj):
                entry[name] = {"type": "function", "params": describe(obj)}
            elif inspect.isclass(obj):
                methods = {}
                for mname, m in sorted(vars(obj).items()):
                    if mname.startswith("_") and mname != "__init__":
                        continue
                    if inspect.isfunction(m):
                        methods[mname] = describe(m)
                entry[name] = {"type": "class", "methods": methods}
        out[modname] = entry
    with open(os.path.join(HERE, "surface_signatures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
