"""Assembles the reference pipeline objects (option dict -> model -> sde) the way trainUM.py:189-217 /
testUM.py:66-96 do, for drivers, bench.py, __graft_entry__.smoke() and tests."""
import copy
import os

import torch

from . import options as option
from .models import create_model
from .models.SDEs import create_sde

DEFAULT_YAML = os.path.join(os.path.dirname(os.path.abspath(__file__)), "Configurations", "um_iddm.yml")


def load_options(path=DEFAULT_YAML, is_train=False):
    return option.dict_to_nonedict(option.parse(path, is_train=is_train))


def build(opt=None, phase="test", device=None, T=None, seed=0, dist=False, sde_overrides=None, score_map_dropout=None, score_map_decoder=None, score_map_if_flash=None):
    """-> (model: CLIPDriftModel, sde).  Random-init weights (seed) as the reference does for a fresh run.
    score_map_dropout: overrides the model option of that name (training-mode dropout of the ScoreMapModule decoder blocks; 0.1).
    score_map_decoder: overrides the model option of that name ("ContextDecoder" | "ContextDecoder_Hierachical").
    score_map_if_flash: overrides the model option of that name (the fp16 form of the Hierachical decoder's attentions)."""
    opt = opt or load_options()
    train_opt = copy.deepcopy(dict(opt['train']))
    train_opt['dist'] = dist
    which = (opt['test'] or {}).get('which_model') if phase == 'test' and opt['test'] else None
    model_opt = opt['models'][which or train_opt['which_model']]
    if score_map_dropout is not None:
        model_opt = copy.copy(model_opt)
        model_opt['score_map_dropout'] = float(score_map_dropout)
    if score_map_decoder is not None:
        model_opt = copy.copy(model_opt)
        model_opt['score_map_decoder'] = str(score_map_decoder)
    if score_map_if_flash is not None:
        model_opt = copy.copy(model_opt)
        model_opt['score_map_if_flash'] = bool(score_map_if_flash)
    torch.manual_seed(seed)
    from .models.drift_noise_model import create_CLIPDriftModel  # registry target, imported for the device kwarg
    model = create_CLIPDriftModel(train_opt, model_opt, phase=phase, device=device) if device is not None else \
        create_model(train_opt, model_opt, phase=phase)
    sde_opt = dict(opt['sdes'][train_opt['which_sde']])
    if T is not None:
        sde_opt['T'] = T
    if sde_overrides:
        sde_opt.update(sde_overrides)
    sde = create_sde(model.get_nets(), sde_opt)
    sde.set_gpu(model.device)
    model.set_sde(sde)
    return model, sde
