"""Registry + build_model with the reference's surface (diffusion/model/builder.py:5-14; mmcv.Registry semantics:
register_module() on classes and factory functions, build(cfg, default_args))."""
from .utils import set_grad_checkpoint


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(obj):
            key = name or obj.__name__
            if key in self._module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._module_dict[key] = obj
            return obj
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        typ = args.pop("type")
        obj = self._module_dict.get(typ) if isinstance(typ, str) else typ
        if obj is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return obj(**args)


MODELS = Registry("models")


def build_model(cfg, use_grad_checkpoint=False, use_fp32_attention=False, gc_step=1, **kwargs):
    if isinstance(cfg, str):
        cfg = dict(type=cfg)
    model = MODELS.build(cfg, default_args=kwargs)
    if use_grad_checkpoint:
        set_grad_checkpoint(model, use_fp32_attention=use_fp32_attention, gc_step=gc_step)
    return model
