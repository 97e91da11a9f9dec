"""forward_step for three-output models: ``(interm_logits, logits, proj_features)`` -- used by the
UPerNet configs (reference: managers/OCRNet_Manager.py:18-56, which passes ``interm_prediction`` so
that TwoScaleLoss can see the auxiliary head)."""
import torch

from ..losses import LossWrapper
from .BaseManager import BaseManager


class OCRNetManager(BaseManager):
    def forward_step(self, img, lbl, **kwargs):
        ret = dict()
        skip_mem_update = kwargs.get('skip_mem_update', False)
        proj_features, interm_output = None, None
        if isinstance(self.loss, LossWrapper):
            lbl = lbl.long()                      # converted once so that prepare() and forward() see one tensor
            if self.return_features and self.model.training:
                self.loss.prepare(lbl, ready_event=kwargs.get('label_ready'))     # see HRNet_Manager.forward_step
            if self.return_features:
                out = self.model(img.float())
                if len(out) == 3:
                    interm_output, output, proj_features = out
                else:
                    output, proj_features = out
                loss = self.loss(output, lbl.long(), interm_prediction=interm_output,
                                 deep_features=proj_features, epoch=self.epoch, skip_mem_update=skip_mem_update)
            else:
                out = self.model(img.float())
                interm_output, output = out if isinstance(out, (tuple, list)) else (None, out)
                loss = self.loss(output, lbl.long(), interm_prediction=interm_output, epoch=self.epoch)
            if 'individual_losses' in kwargs:
                acc = kwargs['individual_losses']
                for key in self.loss.loss_vals:
                    acc[key] += self.loss.loss_vals[key]
                ret['individual_losses'] = acc
        else:
            out = self.model(img.float())
            interm_output, output = out if isinstance(out, (tuple, list)) else (None, out)
            loss = self.loss(output, lbl.long())
        ret.update(output=output, interm_output=interm_output, feats=proj_features, loss=loss)
        if self.empty_cache:
            torch.cuda.empty_cache()
        return ret
