"""CPU: the drop-in surface.  tests/golden/surface_signatures.json was recorded from the reference with
inspect.signature (make_signatures.py); every public function, class and public method it lists must exist under the same
name in the iron_amd mirror of that module with a COMPATIBLE signature -- the reference's parameters first, same names,
same order, same defaults; the mirror may only append defaulted keyword parameters -- unless it is on the short list of
names that are deliberately not part of this build, each with the reason."""
import importlib
import inspect
import json
import os

import pytest

from _util import GOLDEN

MIRROR = {"models.raytracer": "iron_amd.raytracer", "models.renderer_ggx": "iron_amd.renderer_ggx",
          "models.rendering_func": "iron_amd.rendering_func", "models.fields": "iron_amd.fields", "models.embedder": "iron_amd.embedder",
          "models.renderer": "iron_amd.renderer", "models.network_conf": "iron_amd.network_conf"}

# (module, name[, method]) -> why it is not mirrored.  Everything else in the fixture must be there.
NOT_BUILT = {
    ("models.rendering_func", "render_fn"): "dead code in the reference: calls the undefined get_materials_exp (rendering_func.py:85)",
    ("models.network_conf", "render_fn_comp"): "dead code: references undefined names (network_conf.py:767-870)",
    ("models.network_conf", "render_fn_exp"): "dead code: references undefined names (network_conf.py:767-870)",
    ("models.network_conf", "choose_optmizer"): "training orchestration (optimizer choice), SURVEY 2 row 5: out of scope",
    ("models.network_conf", "choose_renderer_func"): "returns the dead render_fn_* closures above",
    ("models.network_conf", "init_outputs"): "helper of the dead render_fn_* closures",
    ("models.fields", "NeRFdual"): "fork experiment for the NIR/RGB dual background, SURVEY 2 row 3: out of scope",
    ("models.renderer_ggx", "calc_dist_params"): "helper of the anisotropic branch CompositeRenderer never takes (has_anisotropic=False); folded into ggx_core.h",
    ("models.renderer_ggx", "fresnel_conductor_exact"): "internal helper of the conductor heads: lives in csrc/ggx_core.h (tested through G9)",
    ("models.renderer_ggx", "fresnel_dielectric"): "internal helper of the dielectric heads: lives in csrc/ggx_core.h (tested through G9)",
}


# CompositeRenderer: forward (what get_materials_comp / render_fn_comp / model_bed.py call) is ONE fused kernel
# (iron_composite_colocated, csrc/ggx_core.h: composite_point); the ~25 per-term helper methods of the fork's principled-BRDF
# experiments that forward calls, or that nothing calls (forward1, forward_ggx, sheen / clearcoat / flatness terms), are not
# separately exposed.
for _m in ("calc_D_Clearcoat", "calc_D_specular", "calc_F_Clearcoat", "calc_G_Clearcoat", "calc_G_specular", "calc_schlick",
           "dielectric_reflection", "diffuse_reflection", "diffuse_reflection_ggx", "flatness_evaluation", "forward1", "forward_ggx",
           "main_dielectric_reflection", "main_metallic_reflection", "main_specular_reflection", "metallic_reflection",
           "principled_fresnel", "schlick_R0_eta", "schlick_weight", "secondary_isotropic_specular_reflection", "select",
           "sheen_evaluation"):
    NOT_BUILT[("models.renderer_ggx", "CompositeRenderer", _m)] = "per-term helper of the fused composite kernel (csrc/ggx_core.h)"


def _fixture():
    return json.load(open(os.path.join(GOLDEN, "surface_signatures.json")))


def _compatible(ref_params, fn):
    """None if compatible, else a description of the first difference."""
    try:
        mine = list(inspect.signature(fn).parameters.values())
    except (TypeError, ValueError) as e:
        return "no signature: %r" % (e,)
    var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in mine)
    mine_named = [p for p in mine if p.kind not in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL)]
    for i, rp in enumerate(ref_params):
        if rp["kind"] in ("VAR_KEYWORD", "VAR_POSITIONAL"):
            continue
        if i >= len(mine_named):
            return "missing parameter %s" % rp["name"] if not var_kw else None
        mp = mine_named[i]
        if mp.name != rp["name"]:
            return "parameter %d is %s, reference has %s" % (i, mp.name, rp["name"])
        ref_has_default = rp["default"] is not None
        mine_has_default = mp.default is not inspect._empty
        if ref_has_default and not mine_has_default:
            return "parameter %s lost its default %s" % (rp["name"], rp["default"])
        if ref_has_default and repr(mp.default) != rp["default"]:
            return "parameter %s default %r, reference %s" % (rp["name"], mp.default, rp["default"])
        if not ref_has_default and mine_has_default and rp["name"] != "self":
            pass  # an added default is compatible
    for mp in mine_named[len([p for p in ref_params if p["kind"] not in ("VAR_KEYWORD", "VAR_POSITIONAL")]):]:
        if mp.default is inspect._empty:
            return "extra parameter %s has no default" % mp.name
    return None


def test_every_reference_name_is_mirrored_with_a_compatible_signature():
    fx = _fixture()
    problems, checked = [], 0
    for ref_mod, entry in fx.items():
        assert "__import_error__" not in entry, (ref_mod, entry)
        mod = importlib.import_module(MIRROR[ref_mod])
        for name, desc in entry.items():
            if (ref_mod, name) in NOT_BUILT:
                continue
            obj = getattr(mod, name, None)
            if obj is None:
                problems.append("%s.%s is missing" % (MIRROR[ref_mod], name))
                continue
            if desc["type"] == "function":
                checked += 1
                why = _compatible(desc["params"], obj)
                if why:
                    problems.append("%s.%s: %s" % (MIRROR[ref_mod], name, why))
            else:
                for mname, params in desc["methods"].items():
                    if (ref_mod, name, mname) in NOT_BUILT or params is None:
                        continue
                    m = getattr(obj, mname, None)
                    if m is None:
                        problems.append("%s.%s.%s is missing" % (MIRROR[ref_mod], name, mname))
                        continue
                    checked += 1
                    why = _compatible(params, m)
                    if why:
                        problems.append("%s.%s.%s: %s" % (MIRROR[ref_mod], name, mname, why))
    assert not problems, "\n" + "\n".join(problems)
    assert checked >= 70, checked


def test_the_not_built_list_only_names_things_the_reference_has():
    fx = _fixture()
    for key in NOT_BUILT:
        entry = fx[key[0]][key[1]]
        if len(key) == 3:
            assert key[2] in entry["methods"], key


def test_install_as_models_serves_the_reference_import_lines():
    """`from models.renderer_ggx import smithG1`-style imports of a reference caller resolve after install_as_models()."""
    import sys
    import iron_amd
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    try:
        iron_amd.install_as_models()
        from models.raytracer import Camera, RayTracer, render_camera, reparam_points, unique  # noqa: F401
        from models.renderer import NeRFRenderer, NeuSRenderer, sample_pdf  # noqa: F401
        from models.renderer_ggx import GGXColocatedRenderer, smithG1  # noqa: F401
        from models.rendering_func import get_materials, get_materials_comp  # noqa: F401
        assert hasattr(NeuSRenderer, "render_core") and hasattr(NeuSRenderer, "render_core_outside")
    finally:
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        sys.modules.update(saved)
