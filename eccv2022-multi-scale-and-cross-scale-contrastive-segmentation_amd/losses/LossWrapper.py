"""Name -> loss registry and weighted sum of component losses: the drop-in boundary of the hot path
(reference: losses/LossWrapper.py:9-103).  Same constructor (the flat ``loss`` config dict with
``losses``, ``device``, ``dataset``, ``experiment`` ...), same ``forward`` signature, same
``loss_vals`` keys (``<name>``, ``<name>_ms{s}``, ``<name>_cs{k}``)."""
from typing import Union

import torch
from torch import nn

from ..utils import DATASETS_INFO
from .TwoScaleLoss import CITYSCAPES_CLASS_WEIGHTS


def _resolve(name):
    import importlib
    pkg = importlib.import_module(__package__)
    if hasattr(pkg, name):
        return getattr(pkg, name)
    raise KeyError(f"loss class '{name}' is not exported by {__package__}")


class LossWrapper(nn.Module):
    def __init__(self, config: dict):
        super().__init__()
        self.config = config
        self.loss_weightings = config['losses']
        self.device = config['device']
        self.dataset = config['dataset']
        self.experiment = config['experiment']
        self.total_loss = None
        self.loss_classes, self.loss_vals = {}, {}
        self.info_string = ''
        names = DATASETS_INFO[self.dataset].CLASS_INFO[self.experiment][1]
        self.ignore_class = (len(names) - 1) if 255 in names else -1
        for loss_class in self.loss_weightings:
            if loss_class == 'CrossEntropyLoss':
                class_weights = None
                if self.dataset == 'CITYSCAPES':
                    class_weights = torch.FloatTensor(CITYSCAPES_CLASS_WEIGHTS).to(self.device)
                loss_fct = nn.CrossEntropyLoss(ignore_index=self.ignore_class, weight=class_weights)
            else:
                loss_fct = _resolve(loss_class)(config)
            self.loss_classes.update({loss_class: loss_fct})
            self.loss_vals.update({loss_class: 0})
            self.info_string += loss_class + ', '
        self.info_string = self.info_string[:-2]
        self.dc_off = True if 'dc_off_at_epoch' in self.config else False

    def prepare(self, labels: torch.Tensor, ready_event=None, loss_list: list = None):
        """Forward the label tensor to components that can start work before the model forward
        (DenseContrastiveLossV2_ms.prepare).  Optional; not part of the reference surface.  ``loss_list``: the list
        forward() will be called with -- a component that is left out of it is not prepared (its preparation would
        consume the step's randperm draws for nothing)."""
        if labels.dtype != torch.int64:       # forward() would see a different (converted) tensor
            return
        for name, fn in self.loss_classes.items():
            if hasattr(fn, 'prepare') and (loss_list is None or name in loss_list):
                fn.prepare(labels, ready_event=ready_event)

    def _zero(self):
        # the reference's torch.tensor(0., device=...) (LossWrapper.py:48) is a pageable host -> device copy, which
        # on ROCm blocks the host until the device has drained (43 ms per step at the benchmark shape: the whole
        # model forward); a fill kernel gives the same scalar asynchronously
        return torch.zeros((), dtype=torch.float, device=self.device)

    def forward(self,
                prediction: torch.Tensor,
                labels: torch.Tensor,
                loss_list: list = None,
                deep_features: Union[torch.Tensor, list] = None,
                interm_prediction: torch.Tensor = None,
                epoch: int = None,
                skip_mem_update: bool = False) -> torch.Tensor:
        self.total_loss = self._zero()
        loss_list = list(self.loss_weightings.keys()) if loss_list is None else loss_list
        lazy = prediction if hasattr(prediction, 'materialize') else None     # models.ops.UpsampledLogits
        if lazy is not None and any(k not in ('CrossEntropyLoss', 'DenseContrastiveLossV2', 'DenseContrastiveLossV2_ms',
                                              'TwoScaleLoss')
                                    for k in self.loss_weightings if k in loss_list):
            prediction = lazy.materialize()           # a component without a fused path wants the full tensor
        for loss_class in self.loss_weightings:
            if loss_class in loss_list:
                if 'DenseContrastive' in loss_class:
                    assert deep_features is not None, \
                        f'for loss_class {loss_class}, deep_features must be tensor (B,H,W,C) instead got {deep_features}'
                fn = self.loss_classes[loss_class]
                if loss_class == 'LovaszSoftmax':
                    if self.dc_off and epoch is not None and epoch < self.config['dc_off_at_epoch']:
                        loss = self._zero()
                    else:
                        loss = fn(prediction, labels)
                elif loss_class == 'TwoScaleLoss':
                    loss = fn(interm_prediction, prediction, labels.long())
                elif loss_class in ('DenseContrastiveLossV2', 'DenseContrastiveLossV2_ms'):
                    loss = fn(labels, deep_features)
                    if isinstance(loss, tuple):     # bare DCV2 configured with cross_scale_contrast
                        loss = loss[0]
                elif loss_class == 'CrossEntropyLoss' and lazy is not None and type(fn) is nn.CrossEntropyLoss \
                        and fn.reduction == 'mean' and fn.label_smoothing == 0.0:
                    # bilinear up-sampling + weighted CE in one kernel, no full-resolution logits (csrc/dcl_upce.hip)
                    loss = lazy.cross_entropy(labels, weight=fn.weight, ignore_index=fn.ignore_index)
                elif loss_class in ('CrossEntropyLoss', 'OhemCrossEntropy'):
                    loss = fn(prediction if lazy is None else lazy.materialize(), labels)
                else:
                    print("Error: Loss class '{}' not recognised!".format(loss_class))
                    loss = self._zero()
            else:
                loss = self._zero()
                if hasattr(self.loss_classes[loss_class], 'discard_prepared'):
                    self.loss_classes[loss_class].discard_prepared()     # prepared for a step that leaves it out

            loss = loss * self.loss_weightings[loss_class]
            self.loss_vals[loss_class] = loss.detach()
            if loss_class == 'DenseContrastiveLossV2_ms' and loss_class in loss_list:
                mod = self.loss_classes[loss_class]
                if hasattr(mod, 'ms_losses'):
                    for scale, loss_val_ms in enumerate(mod.ms_losses):
                        self.loss_vals.update({f'{loss_class}_ms{scale}': loss_val_ms})
                if mod.cross_scale_contrast and hasattr(mod, 'cs_losses'):
                    for cscale, loss_val_cs in enumerate(mod.cs_losses):
                        self.loss_vals.update({f'{loss_class}_cs{cscale}': loss_val_cs})
            self.total_loss = self.total_loss + loss
        return self.total_loss
