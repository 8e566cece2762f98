"""Loader for the reference's Python config files (configs/**/*.py; mmcv `Config` contract).

Supported, as used by the reference (SURVEY.md section 5 "Config / flags"): executing the .py file,
`_base_` (str or list) inheritance with recursive dict merge, `_delete_=True`
(configs/instance/coco_b48n17.py:237), attribute access on nested dicts, `merge_from_dict` with dotted
keys (`--cfg-options k=v`, tools/train.py:81-90).
"""
import copy
import os


class ConfigDict(dict):
    """dict with attribute access (missing attribute -> AttributeError, like mmcv.ConfigDict)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{self.__class__.__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def __deepcopy__(self, memo):
        return ConfigDict({copy.deepcopy(k, memo): copy.deepcopy(v, memo) for k, v in self.items()})


def to_config_dict(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: to_config_dict(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [to_config_dict(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(to_config_dict(v) for v in obj)
    return obj


DELETE_KEY = '_delete_'
BASE_KEY = '_base_'


def _merge_a_into_b(a, b):
    """values of `a` override `b`; dicts merge recursively unless a carries `_delete_=True`."""
    b = copy.deepcopy(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} is a '
                                f'dict in the child config but is of type {type(b[k])} in base config. '
                                f'You may set `{DELETE_KEY}=True` to ignore the base config.')
            b[k] = _merge_a_into_b(v, b[k])
        elif isinstance(v, dict):
            v = copy.deepcopy(v)
            v.pop(DELETE_KEY, None)
            b[k] = v
        else:
            b[k] = copy.deepcopy(v)
    return b


def _load_py(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'config file {filename} does not exist')
    if not filename.endswith('.py'):
        raise IOError('Only py type are supported now!')
    scope = {'__file__': filename, '__name__': '__cgg_config__'}
    with open(filename, 'r', encoding='utf-8') as f:
        code = compile(f.read(), filename, 'exec')
    exec(code, scope)
    cfg = {k: v for k, v in scope.items()
           if not k.startswith('__') and not isinstance(v, type(os)) and not callable(v)}
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, list) else [base]
        merged = {}
        for b in base:
            bcfg = _load_py(os.path.join(os.path.dirname(filename), b))
            dup = merged.keys() & bcfg.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases. Duplicate keys: {dup}')
            merged.update(bcfg)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg


class Config:

    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, but got {type(cfg_dict)}')
        object.__setattr__(self, '_cfg_dict', to_config_dict(cfg_dict))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load_py(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = to_config_dict(value)

    __setitem__ = __setattr__

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """options: {'a.b.c': v}; list elements addressed by integer keys ('pipeline.0.type')."""
        for full_key, v in options.items():
            d = self._cfg_dict
            keys = full_key.split('.')
            for sub in keys[:-1]:
                if isinstance(d, list):
                    d = d[int(sub)]
                else:
                    d = d.setdefault(sub, ConfigDict())
            if isinstance(d, list):
                d[int(keys[-1])] = to_config_dict(v)
            else:
                d[keys[-1]] = to_config_dict(v)


def parse_option_value(v):
    """`--cfg-options key=value` values ([3P] mmcv DictAction): int / float / bool / None, "[a,b]" / "(a,b)" lists and
    comma-separated lists; anything else stays a string."""
    import json
    for cast in (int, float):
        try:
            return cast(v)
        except ValueError:
            pass
    if v in ('True', 'False', 'None'):
        return {'True': True, 'False': False, 'None': None}[v]
    if v.startswith(('[', '(')):
        return json.loads(v.replace('(', '[').replace(')', ']').replace("'", '"'))
    return [parse_option_value(x) for x in v.split(',')] if ',' in v else v
