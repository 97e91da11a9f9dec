"""Lovasz-Softmax loss (Berman et al., CVPR 2018) with the reference's constructor and options
(losses/LovaszSoftmax.py:8-86): ``per_image``, ``classes_to_ignore`` (default: the dataset's ignore id),
``classes_to_consider`` in {'present', 'all', [ids]}.  Plain PyTorch: it is not on the BASELINE hot path
(no shipped config selects it) but LossWrapper dispatches to it by name."""
import torch
import torch.nn as nn

from ..utils import DATASETS_INFO


def lovasz_grad(gt_sorted: torch.Tensor) -> torch.Tensor:
    """Gradient of the Lovasz extension of the Jaccard loss w.r.t. sorted errors (Alg. 1 of the paper)."""
    gts = gt_sorted.sum()
    inter = gts - gt_sorted.cumsum(0)
    union = gts + (1.0 - gt_sorted).cumsum(0)
    jac = 1.0 - inter / union
    if gt_sorted.numel() > 1:
        jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
    return jac


class LovaszSoftmax(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.eps = torch.as_tensor(1e-10)
        self.experiment = config['experiment']
        self.dataset = config['dataset']
        names = DATASETS_INFO[self.dataset].CLASS_INFO[self.experiment][1]
        default_ignore = len(names) - 1 if 255 in names else None
        self.per_image = config.get('per_image', False)
        self.classes_to_ignore = config.get('classes_to_ignore', default_ignore)
        self.classes_to_consider = config.get('classes_to_consider', 'present')

    def _flat(self, prob, lbl):
        c = prob.shape[1]
        prob = prob.permute(0, 2, 3, 1).reshape(-1, c)
        lbl = lbl.reshape(-1)
        if self.classes_to_ignore is None:
            return prob, lbl
        valid = lbl != self.classes_to_ignore
        return prob[valid], lbl[valid]

    def _loss_flat(self, prob, lbl):
        if prob.numel() == 0:
            return prob * 0.0                      # only void pixels: zero loss, zero gradient
        c = prob.shape[1]
        classes = list(range(c)) if self.classes_to_consider in ('all', 'present') else self.classes_to_consider
        terms = []
        for k in classes:
            fg = (lbl == k).float()
            if self.classes_to_consider == 'present' and fg.sum() == 0:
                continue
            err = (fg - prob[:, k]).abs()
            err_sorted, perm = torch.sort(err, 0, descending=True)
            terms.append(torch.dot(err_sorted, lovasz_grad(fg[perm.detach()])))
        if not terms:
            return 0
        return terms[0] if len(terms) == 1 else sum(terms[1:], terms[0]) / len(terms)

    def forward(self, prediction: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        p = torch.softmax(prediction, dim=1)
        if self.per_image:
            per = [self._loss_flat(*self._flat(pi.unsqueeze(0), ti.unsqueeze(0))) for pi, ti in zip(p, target)]
            return per[0] if len(per) == 1 else sum(per[1:], per[0]) / len(per)
        return self._loss_flat(*self._flat(p, target))
