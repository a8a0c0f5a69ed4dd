"""The three helpers of ldm/util.py that the hot path's callers use (exists / default / instantiate_from_config, ldm/util.py:62-89),
plus the batch-size sanity warning the samplers share."""
import importlib
from inspect import isfunction

# reference YAMLs name `ldm.*` / `cldm.*` targets (anydoor.yaml:2,22,40 spells the cldm ones with the collection prefix): resolved inside
# this package so the config files work unchanged
_TARGET_PREFIXES = (("AnyEdit_Collection.other_modules.cldm.", "anyedit_amd.cldm."), ("cldm.", "anyedit_amd.cldm."), ("ldm.", "anyedit_amd.ldm."))
_PLACEHOLDERS = ("__is_first_stage__", "__is_unconditional__")


def exists(x):
    return x is not None


def default(val, d):
    """`val`, or the fallback `d` (called when it is a plain function) if `val` is None."""
    if val is not None:
        return val
    return d() if isfunction(d) else d


def get_obj_from_str(string, reload=False):
    module, _, attr = string.rpartition(".")
    for old, new in _TARGET_PREFIXES:
        if module.startswith(old):
            module = new + module[len(old):]
            break
    return getattr(importlib.import_module(module), attr)


def instantiate_from_config(config):
    """{"target": "pkg.mod.Class", "params": {...}} -> Class(**params); the two placeholder strings give None (ldm/util.py:74-81)."""
    try:
        target = config["target"]
    except (KeyError, TypeError, IndexError):
        if isinstance(config, str) and config in _PLACEHOLDERS:
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(target)(**config.get("params", {}))


def warn_conditioning_batch(conditioning, batch_size):
    """The samplers' only argument check (ddim.py:74-88, dpm_solver/sampler.py:50-57): print — never raise — when a conditioning tensor's
    leading dimension is not the batch size.  Dicts are judged by their first entry (nested lists by their first tensor), lists entry by
    entry."""
    if conditioning is None:
        return
    if isinstance(conditioning, dict):
        probe = next(iter(conditioning.values()))
        while isinstance(probe, list):
            probe = probe[0]
        sizes = [probe.shape[0]]
    elif isinstance(conditioning, list):
        sizes = [c.shape[0] for c in conditioning]
    else:
        sizes = [conditioning.shape[0]]
    for n in sizes:
        if n != batch_size:
            print(f"Warning: Got {n} conditionings but batch-size is {batch_size}")
