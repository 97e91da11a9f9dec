"""Per-step training metrics on the device (reference: utils/torch_utils.py:157-283).

The reference builds two one-hot matrices ([N*H*W, C] int64, ~1 GB at 12x512x1024x20) and multiplies
them; here the confusion matrix is a single (pred, target) 2-D histogram via ``torch.bincount`` --
same integer result, ~50 MB of traffic."""
import torch

from .datasets_info import DATASETS_INFO


@torch.no_grad()
def t_get_confusion_matrix(prediction, target, dataset, existing_matrix=None, no_ignore_class=True):
    """prediction: logits [N, C, H, W]; target: [N, H, W].  Rows = predicted class, cols = target."""
    C = prediction.shape[1]
    pred = prediction.argmax(1).reshape(-1)
    t = target.reshape(-1).to(torch.int64)
    with_ignore = [len(ci[1]) - 1 for ci in DATASETS_INFO[dataset].CLASS_INFO if 255 in ci[1]]
    cols = C + 1 if (no_ignore_class and C in with_ignore) else C
    cm = torch.bincount(pred * cols + t, minlength=C * cols).view(C, cols)[:, :C].to(torch.int)
    if existing_matrix is not None:
        cm = cm + existing_matrix
    return cm


@torch.no_grad()
def t_get_pixel_accuracy(cm):
    diag = torch.diag(cm).float()
    acc = diag.sum() / cm.sum()
    rows = cm.sum(1).float()
    rows[rows == 0] = 1
    return acc, (diag / rows).mean()


@torch.no_grad()
def t_get_mean_iou(cm, calculate_mean=True):
    """IoU per class = diag / (row + col - diag); classes absent from both are skipped in the mean."""
    cm = cm.float()
    diag = torch.diag(cm)
    union = cm.sum(0) + cm.sum(1) - diag
    iou = torch.where(union > 0, diag / union.clamp(min=1), torch.full_like(diag, float('nan')))
    if calculate_mean:
        valid = ~torch.isnan(iou)
        return iou[valid].mean() if valid.any() else torch.tensor(0.0, device=cm.device)
    return iou
