"""SDE registry: `create_sde(nets, sde_opt)` keyed by sde_opt['class_name'] (trainUM.py:215-216,
testUM.py:89-92; Configurations/config.yml:169-175)."""
import importlib


def create_sde(nets, sde_opt):
    sde_opt = dict(sde_opt)
    class_name = sde_opt.pop("class_name")
    module = importlib.import_module(f"{__package__}.{class_name}")
    cls = getattr(module, class_name)
    return cls(nets=nets, **sde_opt)
