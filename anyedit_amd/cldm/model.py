"""Mirror of AnyEdit_Collection/other_modules/cldm/model.py (:8-28): checkpoint / config loading for the ControlLDM (AnyDoor) path —
`create_model(config_path)` + `model.load_state_dict(load_state_dict(ckpt, location))` is how visual_reference_tool.py:370-375 builds it.

On-disk formats: a `.safetensors` file is a flat name -> tensor map; a `.ckpt` / `.pth` is a torch pickle that is either that map or a
dict holding it under 'state_dict' (Lightning checkpoints).  YAML configs are read with PyYAML (the reference uses OmegaConf only as a
YAML reader here); `target:` strings naming the reference's `ldm.*` / `cldm.*` classes resolve to this package's mirrors.
"""
import os

import torch

from anyedit_amd.ldm.util import instantiate_from_config


def get_state_dict(d):
    return d.get('state_dict', d)


def load_state_dict(ckpt_path, location='cpu'):
    _, extension = os.path.splitext(ckpt_path)
    if extension.lower() == ".safetensors":
        import safetensors.torch
        state_dict = safetensors.torch.load_file(ckpt_path, device=location)
    else:
        state_dict = get_state_dict(torch.load(ckpt_path, map_location=torch.device(location)))
    state_dict = get_state_dict(state_dict)
    print(f'Loaded state_dict from [{ckpt_path}]')
    return state_dict


def create_model(config_path):
    import yaml
    with open(config_path, "r") as f:
        config = yaml.safe_load(f)
    model = instantiate_from_config(config["model"]).cpu()
    print(f'Loaded model config from [{config_path}]')
    return model
