"""Mirror of ldm/util.py helpers used on the hot path (ldm/util.py:74-89)."""
import importlib
from inspect import isfunction


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    # reference configs name `ldm.*` targets; resolve them to this package so YAMLs work unchanged
    if module.startswith("ldm."):
        module = "anyedit_amd." + module
    elif module.startswith("AnyEdit_Collection.other_modules.cldm."):     # anydoor.yaml:2,22,40 name the cldm classes this way
        module = "anyedit_amd.cldm." + module[len("AnyEdit_Collection.other_modules.cldm."):]
    elif module.startswith("cldm."):
        module = "anyedit_amd." + module
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    """ldm/util.py:74-81: {"target": "pkg.mod.Class", "params": {...}} reflection."""
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))
