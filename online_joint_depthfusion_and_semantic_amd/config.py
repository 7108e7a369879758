"""Attribute-style config dictionaries (the reference uses EasyDict, utils/loading.py:9-19) and the
default hot-path configuration (keys of configs/fusion/replica_accuracy.yaml)."""
import copy

import yaml


class AttrDict(dict):
    """dict with attribute access, recursively (EasyDict-compatible for the keys the path reads)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def load_config_from_yaml(path):
    with open(path) as f:
        return AttrDict(yaml.safe_load(f))


def default_config(h=240, w=320, semantics=False, use_semantics=None, n_classes=30, model='v3',
                   depth_key='tof_depth', integrate_mode='fast', init_value=0.1):
    """Config with the hot-path keys (SETTINGS / FUSION_MODEL / SEMANTIC_2D_MODEL / DATA)."""
    if use_semantics is None:
        use_semantics = bool(semantics)
    return AttrDict({
        'SETTINGS': {'gpu': True, 'implementation': 'efficient', 'device': 'cuda:0', 'seed': 1911,
                     'integrate_mode': integrate_mode},
        'FUSION_MODEL': {'name': model, 'output_scale': 1.0, 'n_points': 9, 'n_tail_points': 7,
                         'growth_factor': 6, 'use_semantics': bool(use_semantics),
                         'arithmetic': 'f16x3'},  # 'f16x3' | 'f32' (include/ojf.h OJF_ARITH_*)
        'SEMANTIC_2D_MODEL': {'stage': 2, 'n_classes': n_classes},
        'TRAINING': {'optimization': {'accumulation_steps': 8, 'clipping': True}},
        'TESTING': {'outlier_filter_val': 2},
        'DATA': {'semantics': 'class{}'.format(n_classes) if semantics else None,
                 'semantic_strategy': 'gt', 'semantic_grid': bool(semantics), 'input': depth_key,
                 'target': 'depth_gt', 'resx': w, 'resy': h, 'init_value': init_value, 'pad': 0},
    })


def database_config(config):
    """The flat config ``Database`` reads (utils/setup.py:80-85 + get_data_config :20-70)."""
    d = AttrDict(copy.deepcopy(dict(config.DATA)))
    d.device = config.SETTINGS.device
    d.implementation = config.SETTINGS.implementation
    d.transform = None
    d.n_classes = config.SEMANTIC_2D_MODEL.n_classes if config.DATA.semantics else None
    return d
