"""Checkpoint and config loading for the ControlLDM (AnyDoor) path — same three entry points, names and return values as
AnyEdit_Collection/other_modules/cldm/model.py (:8-28); visual_reference_tool.py:370-375 calls `create_model(cfg)` followed by
`model.load_state_dict(load_state_dict(ckpt, location))`.

On-disk formats handled: `.safetensors` (flat name -> tensor map) and torch pickles (`.ckpt` / `.pth` / `.bin`) holding either the
flat map or a Lightning dict with the map under 'state_dict'.  YAML is read with PyYAML (the reference uses OmegaConf only as a YAML
reader here); `target:` strings that name the reference's `ldm.*` / `cldm.*` classes resolve to this package's mirrors.
"""
from pathlib import Path

import torch

from anyedit_amd.ldm.util import instantiate_from_config


def get_state_dict(d):
    """Unwrap a Lightning-style checkpoint: the tensors live under 'state_dict' when that key exists."""
    return d["state_dict"] if "state_dict" in d else d


def _read_safetensors(path, location):
    from safetensors.torch import load_file
    return load_file(str(path), device=location)


def trusted_torch_load(path, location="cpu"):
    """torch.load for reference checkpoints.  torch >= 2.6 unpickles with weights_only=True by default, which rejects Lightning
    `.ckpt` files (callbacks, hyper_parameters, OmegaConf nodes next to the tensors) that the reference loads on its older torch.
    The safe loader is tried first; only when it refuses AND the caller opted in (ANYEDIT_TRUST_CHECKPOINTS=1 — unpickling executes
    code from the file) is the full unpickler used."""
    import os
    import pickle
    try:
        return torch.load(str(path), map_location=torch.device(location), weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        if os.environ.get("ANYEDIT_TRUST_CHECKPOINTS") != "1":
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({str(e).splitlines()[0]}). Checkpoints that carry non-tensor objects "
                               "(Lightning .ckpt) need the full unpickler: set ANYEDIT_TRUST_CHECKPOINTS=1 if you trust this file.") from e
        return torch.load(str(path), map_location=torch.device(location), weights_only=False)


def _read_pickle(path, location):
    return trusted_torch_load(path, location)


def load_state_dict(ckpt_path, location='cpu'):
    path = Path(ckpt_path)
    reader = _read_safetensors if path.suffix.lower() == ".safetensors" else _read_pickle
    state_dict = get_state_dict(get_state_dict(reader(path, location)))   # the reference unwraps twice as well (model.py:18-19)
    print(f'Loaded state_dict from [{ckpt_path}]')
    return state_dict


def create_model(config_path):
    import yaml
    config = yaml.safe_load(Path(config_path).read_text())
    model = instantiate_from_config(config["model"]).cpu()
    print(f'Loaded model config from [{config_path}]')
    return model
