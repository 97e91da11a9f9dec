"""Auxiliary-head + final-head loss pair (reference: losses/TwoScaleLoss.py:9-68): two losses of
the same kind applied to the intermediate and the final logits, weighted 0.4 / 1.0 by default.
Thin PyTorch wrapper (HBM-bound CE over logits; SURVEY.md section 8 row f1 is the planned fusion)."""
import torch
import torch.nn as nn
from torch.nn import CrossEntropyLoss

from ..utils import DATASETS_INFO

CITYSCAPES_CLASS_WEIGHTS = [0.8373, 0.918, 0.866, 1.0345, 1.0166, 0.9969, 0.9754, 1.0489, 0.8786,
                            1.0023, 0.9539, 0.9843, 1.1116, 0.9037, 1.0865, 1.0955, 1.0865, 1.1529,
                            1.0507]


def _resolve(name):
    from . import __dict__ as registry
    if name == 'CrossEntropyLoss':
        return CrossEntropyLoss
    if name in registry:
        return registry[name]
    raise KeyError(name)


class TwoScaleLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        interm_loss_class = _resolve(config['interm']['name'])
        final_loss_class = _resolve(config['final']['name'])
        self.w_interm = config['interm']['weight'] if 'weight' in config['interm'] else 0.4
        self.w_final = config['final']['weight'] if 'weight' in config['final'] else 1.0
        self.ignore_label = -100
        self.dataset = config['dataset']
        self.experiment = config['experiment']
        names = DATASETS_INFO[self.dataset].CLASS_INFO[self.experiment][1]
        self.ignore_label = len(names) - 1 if 255 in names.keys() else len(names)
        config['interm'].update({"experiment": config['experiment'], "dataset": self.dataset})
        config['final'].update({"experiment": config['experiment'], "dataset": self.dataset})
        if config['interm']['name'] == 'CrossEntropyLoss' and config['final']['name'] == 'CrossEntropyLoss':
            class_weights = None
            if self.dataset == 'CITYSCAPES':
                class_weights = torch.FloatTensor(CITYSCAPES_CLASS_WEIGHTS)
                if 'device' in config:
                    class_weights = class_weights.to(config['device'])
            self.loss_interm = interm_loss_class(*config['interm']['args'], ignore_index=self.ignore_label,
                                                 weight=class_weights)
            self.loss_final = final_loss_class(*config['final']['args'], ignore_index=self.ignore_label,
                                               weight=class_weights)
        elif config['interm']['name'] == config['final']['name']:
            self.loss_interm = interm_loss_class(config['interm'])
            self.loss_final = final_loss_class(config['final'])
        else:
            raise NotImplementedError('different losses for interm {} and final {}'.format(
                config['interm'], config['final']))

    @staticmethod
    def _apply(fn, logits, target):
        """``fn(logits, target)``; logits kept at low resolution (models.ops.UpsampledLogits) go through the fused
        up-sampling + cross-entropy kernel when fn is a plain mean-reduced nn.CrossEntropyLoss, else are materialised."""
        if hasattr(logits, 'materialize'):
            if type(fn) is nn.CrossEntropyLoss and fn.reduction == 'mean' and fn.label_smoothing == 0.0:
                return logits.cross_entropy(target, weight=fn.weight, ignore_index=fn.ignore_index)
            logits = logits.materialize()
        return fn(logits, target)

    def forward(self, logits_interm, logits_final, target):
        loss_final = self._apply(self.loss_final, logits_final, target)
        loss_interm = self._apply(self.loss_interm, logits_interm, target)
        return loss_final * self.w_final + loss_interm * self.w_interm
