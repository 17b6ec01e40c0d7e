"""Network registry, same pattern as the reference's models/__init__.py:4-12 one level down:
`create_net(settings, CLIP_ScoreMapModule=...)` keyed by settings['module_name'] / ['class_name']
(models/drift_noise_model.py:142-143; Configurations/config.yml:106-131)."""
import importlib


def create_net(settings, CLIP_ScoreMapModule=None):
    settings = dict(settings)
    module_name = settings.pop("module_name")
    class_name = settings.pop("class_name")
    module = importlib.import_module(f"{__package__}.{module_name}")
    cls = getattr(module, class_name)
    return cls(CLIP_ScoreMapModule=CLIP_ScoreMapModule, **settings)
