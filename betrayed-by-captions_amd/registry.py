"""Registries with the `register_module()` / `build()` contract of the reference's plug-in API.

The reference registers its classes into mmcv/mmdet registries (`@HEADS.register_module()` at
open_set/models/mask2former_head.py:33, `@DETECTORS.register_module()` at mask2former.py:6,
`@LOSSES.register_module()` at losses/grounding_loss.py:79, `@BBOX_ASSIGNERS.register_module()` at
assigners/mask_hungarian_assigner.py:14) and instantiates everything from config dicts through the
`type=` key. This module provides the same contract so the reference's configs drive this package
unchanged: `REG.register_module()` as decorator (optionally `name=`, `force=`), `REG.build(cfg)` /
`build_from_cfg(cfg, REG, default_args)`: pops `type`, the remaining keys become constructor kwargs;
an unknown `type` raises KeyError naming the registry, a non-dict cfg raises TypeError.
"""
import copy
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    args = copy.copy(dict(cfg))
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type) or inspect.isfunction(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f'{obj_cls.__name__}: {e}')


class Registry:

    def __init__(self, name, build_func=None):
        self._name = name
        self._module_dict = {}
        self.build_func = build_func or build_from_cfg

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, module, name=None, force=False):
        if not (inspect.isclass(module) or inspect.isfunction(module)):
            raise TypeError(f'module must be a class or a function, but got {type(module)}')
        names = [module.__name__] if name is None else ([name] if isinstance(name, str) else list(name))
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
            self._module_dict[n] = module

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if module is not None:
            self._register(module, name, force)
            return module

        def _deco(cls):
            self._register(cls, name, force)
            return cls

        return _deco


# The registries the reference (and the [3P] `type=` strings of its configs) resolve through.
DETECTORS = Registry('detector')
BACKBONES = Registry('backbone')
NECKS = Registry('neck')
HEADS = Registry('head')
LOSSES = Registry('loss')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
MATCH_COST = Registry('match_cost')
PLUGIN_LAYERS = Registry('plugin layer')
ATTENTION = Registry('attention')
FEEDFORWARD_NETWORK = Registry('feed-forward network')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
POSITIONAL_ENCODING = Registry('position encoding')
DATASETS = Registry('dataset')
PIPELINES = Registry('pipeline')


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet.models.build_detector (tools/train.py:220, tools/test.py:236)."""
    if train_cfg is not None or test_cfg is not None:
        import warnings
        warnings.warn('train_cfg and test_cfg is deprecated, please specify them in model', UserWarning)
    assert cfg.get('train_cfg') is None or train_cfg is None, \
        'train_cfg specified in both outer field and model field '
    assert cfg.get('test_cfg') is None or test_cfg is None, \
        'test_cfg specified in both outer field and model field '
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_assigner(cfg, **default_args):
    return BBOX_ASSIGNERS.build(cfg, default_args=default_args or None)


def build_sampler(cfg, **default_args):
    return BBOX_SAMPLERS.build(cfg, default_args=default_args or None)


def build_match_cost(cfg, default_args=None):
    return MATCH_COST.build(cfg, default_args=default_args)


def build_plugin_layer(cfg, postfix='', **kwargs):
    """mmcv.cnn.build_plugin_layer: returns (name, layer) (open_set/models/mask2former_head.py:117)."""
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict')
    if 'type' not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    cls = PLUGIN_LAYERS.get(layer_type)
    if cls is None:
        raise KeyError(f'Unrecognized plugin type {layer_type}')
    abbr = getattr(cls, '_abbr_', layer_type.lower())
    return abbr + str(postfix), cls(**kwargs, **cfg_)


def build_attention(cfg, default_args=None):
    return ATTENTION.build(cfg, default_args=default_args)


def build_feedforward_network(cfg, default_args=None):
    return FEEDFORWARD_NETWORK.build(cfg, default_args=default_args)


def build_transformer_layer(cfg, default_args=None):
    return TRANSFORMER_LAYER.build(cfg, default_args=default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return TRANSFORMER_LAYER_SEQUENCE.build(cfg, default_args=default_args)


def build_positional_encoding(cfg, default_args=None):
    return POSITIONAL_ENCODING.build(cfg, default_args=default_args)
