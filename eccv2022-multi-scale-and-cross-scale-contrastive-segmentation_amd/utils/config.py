"""JSON config handling with the reference's schema (utils/config_parsers.py:10-80,
utils/defaults.py:179-251, main.py:86-149): defaults merged under the file's values, dataset /
experiment copied into the ``graph`` and ``loss`` blocks, and the ``--loss ms|ms_cs|ce`` rewrite."""
import copy
import json

DEFAULT_CONFIG_DICT = {'mode': 'training', 'debugging': False, 'log_every_n_epochs': 100, 'max_valid_imgs': 10,
                       'cuda': True, 'gpu_device': [0], 'parallel': False, 'seed': 0, 'tta': False,
                       'valid_freq': 1, 'empty_cache': False}
DEFAULT_CONFIG_NESTED_DICT = {
    'data': {'split': 1, 'batch_size': 10, 'num_workers': 0, 'experiment': 1},
    'train': {'epochs': 50, 'lr_fct': 'exponential', 'lr_batchwise': False, 'lr_restarts': [],
              'lr_restart_vals': 1, 'lr_params': None, 'learning_rate': 0.01},
    'loss': {'temperature': 0.1, 'dominant_mode': 'all', 'label_scaling_mode': 'nn'},
}


def merge_defaults(config: dict) -> dict:
    config = copy.deepcopy(config)
    for k, v in DEFAULT_CONFIG_DICT.items():
        config.setdefault(k, copy.deepcopy(v))
    for block, defaults in DEFAULT_CONFIG_NESTED_DICT.items():
        if block in config:
            for k, v in defaults.items():
                config[block].setdefault(k, copy.deepcopy(v))
    if isinstance(config['gpu_device'], int):      # the reference crashes on an int here (LoggingManager.py:117)
        config['gpu_device'] = [config['gpu_device']]
    dataset = config['data']['dataset']
    config.setdefault('graph', {})['dataset'] = dataset
    if 'loss' in config:
        config['loss']['dataset'] = dataset
        config['loss']['experiment'] = config['data']['experiment']
    return config


def parse_config(path_or_dict) -> dict:
    if isinstance(path_or_dict, dict):
        return merge_defaults(path_or_dict)
    with open(path_or_dict) as f:
        return merge_defaults(json.load(f))


def apply_loss_switch(config: dict, mode: str) -> dict:
    """main.py:97-113: ``ms`` / ``ms_cs`` select CE + 0.1 * DCV2_ms over a 4-scale projector
    (cross-scale terms only for ``ms_cs``); ``ce`` drops the projector."""
    if mode in ('ms', 'ms_cs'):
        config['loss']['losses'] = {'CrossEntropyLoss': 1, 'DenseContrastiveLossV2_ms': 0.1}
        config['loss']['cross_scale_contrast'] = mode == 'ms_cs'
        config['loss'].setdefault('scales', 4)
        config['loss'].setdefault('weights', [1, 0.7, 0.4, 0.1])
        mp = {'mlp': [[1, -1, 1]], 'scales': 4, 'd': 256, 'use_bn': True, 'before_context': False}
        if config['graph']['model'] == 'UPerNet':
            mp['position'] = 'backbone'
        config['graph'].pop('projector', None)
        config['graph']['ms_projector'] = mp
    elif mode == 'ce':
        config['loss']['losses'] = {'CrossEntropyLoss': 1}
        config['graph'].pop('projector', None)
        config['graph'].pop('ms_projector', None)
    return config
