"""Model registry with the reference's semantics (models/__init__.py:4-12): import
`<package>.<module_name>` and call `create_<class_name>(train_opt, model_opt, phase=...)`."""
import importlib


def create_model(train_opt, model_opt, phase='train'):
    module_name = model_opt['module_name']
    class_name = model_opt['class_name']
    module = importlib.import_module(f'{__package__}.{module_name}')
    create_fn = getattr(module, 'create_' + class_name)
    return create_fn(train_opt, model_opt, phase=phase)
