"""forward_step for two-output models: ``(logits, proj_features)`` (reference:
managers/HRNet_Manager.py:18-54)."""
import torch

from ..losses import LossWrapper
from .BaseManager import BaseManager


class HRNetManager(BaseManager):
    def forward_step(self, img, lbl, **kwargs):
        ret = dict()
        skip_mem_update = kwargs.get('skip_mem_update', False)
        proj_features = None
        if isinstance(self.loss, LossWrapper):
            lbl = lbl.long()                      # converted once so that prepare() and forward() see one tensor
            if self.return_features and self.model.training:
                # label stage on a side stream: overlaps the model forward and, given the label's ready event,
                # the tail of the previous step as well
                self.loss.prepare(lbl, ready_event=kwargs.get('label_ready'))
            if self.return_features:
                output, proj_features = self.model(img.float())
                loss = self.loss(output, lbl.long(), deep_features=proj_features, epoch=self.epoch,
                                 skip_mem_update=skip_mem_update)
            else:
                output = self.model(img.float())
                loss = self.loss(output, lbl.long(), epoch=self.epoch)
            if 'individual_losses' in kwargs:
                acc = kwargs['individual_losses']
                for key in self.loss.loss_vals:
                    acc[key] += self.loss.loss_vals[key]
                ret['individual_losses'] = acc
        else:
            output = self.model(img.float())
            loss = self.loss(output, lbl.long())
        ret.update(output=output, interm_output=None, feats=proj_features, loss=loss)
        if self.empty_cache:
            torch.cuda.empty_cache()
        return ret
