"""Per-step training metrics (reference: utils/torch_utils.py:157-283).

``t_get_confusion_matrix`` keeps the reference's signature and integer result.  On the GPU it is ONE pass over the
logits through ``dcl_confusion_matrix`` (csrc/dcl_metrics.hip: argmax + 2-D histogram, HBM-bound, bit-exact) instead
of the reference's transpose copy + argmax + two one-hot matrices (~1 GB at 12 x 512 x 1024 x 20) + float matmul;
CPU tensors (validation on CPU ranks, tests) take a ``bincount`` formulation of the same histogram.
Targets outside the class range -- on which the reference's ``F.one_hot`` raises -- are counted in
``out_of_range()`` (a device scalar the managers fold into their single per-step D2H) instead of syncing here."""
import ctypes

import torch

from .datasets_info import DATASETS_INFO

_OOB = {}


def _oob_counter(device):
    key = (device.type, device.index)
    if key not in _OOB:
        _OOB[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _OOB[key]


def out_of_range(device):
    """int32 [1] device counter of targets outside the class range seen by t_get_confusion_matrix since the last
    ``take_out_of_range`` (the fused up-sampling + CE kernels treat such targets as ignored; the confusion-matrix
    kernel of the same step is what counts them)."""
    return _oob_counter(torch.device(device))


def take_out_of_range(device):
    """The counter's current value as a new float32 [1] device tensor, and the counter reset to zero (both
    stream-ordered, no host sync): one bad batch is reported once, not by every later flush."""
    c = _oob_counter(torch.device(device))
    v = c.float()
    c.zero_()
    return v


def _cols(C, dataset, no_ignore_class):
    with_ignore = [len(ci[1]) - 1 for ci in DATASETS_INFO[dataset].CLASS_INFO if 255 in ci[1]]
    return C + 1 if (no_ignore_class and C in with_ignore) else C       # torch_utils.py:168-177


@torch.no_grad()
def t_get_confusion_matrix(prediction, target, dataset, existing_matrix=None, no_ignore_class=True):
    """prediction: logits [N, C, H, W]; target: [N, H, W] class ids.  int32 [C, C]: rows = predicted class,
    columns = target class; pixels whose target is the dataset's ignore id (== C) are dropped."""
    C = prediction.shape[1]
    cols = _cols(C, dataset, no_ignore_class)
    if hasattr(prediction, 'materialize'):          # models.ops.UpsampledLogits: logits kept at 1/4 resolution
        pred = getattr(prediction, 'pred', None)
        if pred is not None and pred.is_cuda:       # arg-max map left behind by the fused up-sampling + CE forward
            from .. import _lib
            t = target if target.dtype in (torch.int64, torch.int32, torch.uint8) else target.to(torch.int64)
            t = t.contiguous()
            assert t.numel() == pred.numel(), "target must be [N, H, W]"
            cm = torch.zeros((C, cols), dtype=torch.int32, device=pred.device)
            _lib.check(_lib.lib().dcl_confusion_matrix_pred(_lib.ptr(pred), pred.numel(), _lib.ptr(t),
                                                            t.element_size(), C, cols, _lib.ptr(cm),
                                                            _lib.ptr(_oob_counter(pred.device)),
                                                            _lib.stream_ptr(pred.device)), "dcl_confusion_matrix_pred")
            cm = cm[:, :C]
            return cm + existing_matrix if existing_matrix is not None else cm
        prediction = prediction.materialize().detach()
    if prediction.is_cuda:
        from .. import _lib
        L = _lib.lib()                                  # raises if libdcl_hip.so is missing: no silent fallback
        p = prediction if prediction.dtype == torch.float32 else prediction.float()
        p = p.contiguous()
        t = target
        if t.dtype not in (torch.int64, torch.int32, torch.uint8):
            t = t.to(torch.int64)
        t = t.contiguous()
        assert t.numel() == p.shape[0] * p.shape[2] * p.shape[3], "target must be [N, H, W]"
        cm = torch.zeros((C, cols), dtype=torch.int32, device=p.device)
        st = _lib.stream_ptr(p.device)
        _lib.check(L.dcl_confusion_matrix(_lib.ptr(p), p.shape[0], C, p.shape[2] * p.shape[3], _lib.ptr(t),
                                          t.element_size(), cols, _lib.ptr(cm), _lib.ptr(_oob_counter(p.device)), st),
                   "dcl_confusion_matrix")
        cm = cm[:, :C]
    else:
        pred = prediction.argmax(1).reshape(-1)
        t = target.reshape(-1).to(torch.int64)
        if t.numel() and (int(t.min()) < 0 or int(t.max()) >= cols):
            raise RuntimeError("Class values must be smaller than num_classes.")        # F.one_hot's error
        cm = torch.bincount(pred * cols + t, minlength=C * cols).view(C, cols)[:, :C].to(torch.int)
    if existing_matrix is not None:
        cm = cm + existing_matrix
    return cm


@torch.no_grad()
def t_get_pixel_accuracy(confusion_matrix):
    """(overall pixel accuracy, mean per-predicted-class accuracy), torch_utils.py:201-213."""
    diag = torch.diag(confusion_matrix).to(torch.float)
    acc = torch.sum(diag) / torch.sum(confusion_matrix)
    rows = torch.sum(confusion_matrix, dim=1, dtype=torch.float)
    rows[rows == 0] = 1
    return acc, torch.mean(diag / rows)


@torch.no_grad()
def t_get_miou(confusion_matrix, experiment=None, dataset=None, indices=None, calculate_mean=None):
    """torch_utils.py:253-283: IoU = diag / (row + col - diag) over ``indices`` (default: every non-ignore class);
    a class absent from prediction AND target has IoU NaN -> 0 and STAYS in the mean."""
    calculate_mean = True if calculate_mean is None else calculate_mean
    if indices is None:
        if dataset is not None and experiment is not None:
            indices = [c for c in DATASETS_INFO[dataset].CLASS_INFO[experiment][1].keys() if not c == 255]
        else:
            indices = list(range(confusion_matrix.shape[0]))
    else:
        indices = [c for c in indices if not c == 255]
    diag = confusion_matrix.diag()[indices].to(torch.float)
    row_sum = torch.sum(confusion_matrix, dim=0, dtype=torch.float)[indices]
    col_sum = torch.sum(confusion_matrix, dim=1, dtype=torch.float)[indices]
    iou = diag / (row_sum + col_sum - diag)
    iou[iou != iou] = 0
    return iou.mean() if calculate_mean else iou


@torch.no_grad()
def t_get_mean_iou(confusion_matrix, experiment=None, dataset=None, categories=False, single_class=None,
                   calculate_mean=None, rare=False):
    """With ``experiment`` and ``dataset`` (the reference's call, torch_utils.py:216-250): dict with ``mean_iou``
    (+ ``per_class_iou`` and ``categories`` when asked).  Called with the matrix alone (this repo's managers): the
    mean IoU tensor itself."""
    if experiment is None or dataset is None:
        return t_get_miou(confusion_matrix, calculate_mean=calculate_mean)
    assert experiment in [1, 2, 3], 'experiment must be in [1,2,3] instead got [{}]'.format(experiment)
    assert single_class is None, 'single-class IoU is not part of the hot path'
    mious = {'mean_iou': t_get_miou(confusion_matrix, experiment, dataset, calculate_mean=calculate_mean)}
    if categories:
        mious['per_class_iou'] = t_get_miou(confusion_matrix, experiment, dataset, calculate_mean=False)
        mious['categories'] = {}
        cats = DATASETS_INFO[dataset].CLASS_INFO[experiment][2]
        for categ in cats:
            mious['categories'][categ] = t_get_miou(confusion_matrix, experiment, dataset, indices=cats[categ],
                                                    calculate_mean=calculate_mean)
    return mious


@torch.no_grad()
def t_metrics_from_confusion_matrix(confusion_matrix):
    """(pa, pac, miou) as three 0-dim tensors -- ``t_get_pixel_accuracy`` + ``t_get_mean_iou`` (all classes) of the
    reference in one HIP launch for int32 CUDA matrices (csrc/dcl_metrics.hip: the torch formulation is ~25 tiny
    kernels per training step); CPU / other dtypes take the torch path."""
    cm = confusion_matrix
    if cm.is_cuda and cm.dtype == torch.int32 and cm.dim() == 2 and cm.shape[0] == cm.shape[1] \
            and cm.shape[0] <= 256 and cm.stride(1) == 1:
        from .. import _lib
        out = torch.empty(3, dtype=torch.float32, device=cm.device)
        _lib.check(_lib.lib().dcl_metrics_from_cm(_lib.ptr(cm), cm.shape[0], cm.stride(0), _lib.ptr(out),
                                                  _lib.stream_ptr(cm.device)), "dcl_metrics_from_cm")
        return out[0], out[1], out[2]
    pa, pac = t_get_pixel_accuracy(cm)
    return pa, pac, t_get_miou(cm)
