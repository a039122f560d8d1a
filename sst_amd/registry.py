"""Minimal mmcv-style registry: the plugin API the hot path sits behind.

Reference: mmdet3d/models/builder.py:5-98 — ``MODELS`` (aliased VOXEL_ENCODERS / MIDDLE_ENCODERS) and
mmdet's ``BACKBONES``; configs select classes with ``type='...'`` strings and pass the remaining keys
as constructor kwargs.  The same strings ('DynamicVFE', 'DynamicScatterVFE', 'SIRLayer',
'SSTInputLayerV2', 'SSTv2', 'SIR') resolve here, so configs/sst_refactor and configs/fsd dicts build
unchanged.
"""


class Registry(object):

    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls

        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with the key "type", got {cfg!r}')
        args = dict(cfg)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(f'{obj_type} is not in the {self.name} registry')
        else:
            cls = obj_type
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        return cls(**args)


MODELS = Registry('models')
VOXEL_ENCODERS = MODELS
MIDDLE_ENCODERS = MODELS
BACKBONES = Registry('backbone')
ROI_EXTRACTORS = Registry('roi_extractor')  # mmdet.models.builder.ROI_EXTRACTORS


def build_voxel_encoder(cfg):
    return VOXEL_ENCODERS.build(cfg)


def build_middle_encoder(cfg):
    return MIDDLE_ENCODERS.build(cfg)


def build_backbone(cfg):
    return BACKBONES.build(cfg)
