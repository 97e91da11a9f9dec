"""Single-scale supervised dense pixel-contrastive loss -- drop-in for the reference class of the
same name (losses/DenseContrastiveLossV2.py:11-206): same constructor (flat ``loss`` config dict),
same attributes, same forward signature and return arity; the work runs in libdcl_hip.so."""
import torch
import torch.nn as nn

from ..utils import DATASETS_INFO, printlog
from .engine import EngineConfig, dense_contrast_terms
from .. import _lib
from .engine import _stream_ptr


class _RawBank(torch.autograd.Function):
    """sampled_features[t, c, v] = features[b_t, c, pix[t, v]] in the reference's [T, C, V] layout
    (DenseContrastiveLossV2.py:113-123), differentiable (scatter in backward)."""

    @staticmethod
    def forward(ctx, feat, sc):
        p = sc.plan
        X = torch.empty((p.T, sc.C, p.V), dtype=torch.float32, device=feat.device)
        sn, scs, sp = sc.strides
        _lib.check(_lib.lib().dcl_gather_raw(_lib.ptr(feat), sn, scs, sp, sc.C, _lib.ptr(sc.pix),
                                             _lib.ptr(sc.pair_b), p.T, p.V, _lib.ptr(X),
                                             _stream_ptr()), "dcl_gather_raw")
        ctx.sc = sc
        ctx.meta = (tuple(feat.shape), tuple(feat.stride()))
        return X

    @staticmethod
    def backward(ctx, dX):
        sc = ctx.sc
        shape, strides = ctx.meta
        d = torch.empty_strided(shape, strides, dtype=torch.float32, device=dX.device).zero_()
        sn, scs, sp = sc.strides
        p = sc.plan
        _lib.check(_lib.lib().dcl_scatter_raw(_lib.ptr(dX.float().contiguous()), sn, scs, sp, sc.C, _lib.ptr(sc.pix),
                                              _lib.ptr(sc.pair_b), p.T, p.V, _lib.ptr(d), _stream_ptr()),
                   "dcl_scatter_raw")
        return d, None


class DenseContrastiveLossV2(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.experiment = config['experiment']
        self.dataset = config['dataset']
        names = DATASETS_INFO[self.dataset].CLASS_INFO[self.experiment][1]
        self.num_all_classes = len(names)                                        # ref :16
        self.num_real_classes = self.num_all_classes - 1 if 255 in names else self.num_all_classes
        self.ignore_class = (len(names) - 1) if 255 in names else -1
        self.temperature = config['temperature'] if 'temperature' in config else 0.5
        self.base_temperature = 1.0
        self.min_views_per_class = config['min_views_per_class'] if 'min_views_per_class' in config else 5
        self.label_scaling_mode = config['label_scaling_mode'] if 'label_scaling_mode' in config else 'nn'
        self.cross_scale_contrast = config['cross_scale_contrast'] if 'cross_scale_contrast' in config else False
        self.dominant_mode = 'all'
        self.eps = torch.tensor(1e-10)
        self.metadata = {name: (0.0, 0.0) for name in names.values()}
        self.max_views_per_class = config['max_views_per_class'] if 'max_views_per_class' in config else 2500
        self.max_features_total = config['max_features_total'] if 'max_features_total' in config else 10000
        self.log_this_step = False
        self._scale = None
        self.last_state = None            # StepState of the most recent forward (plans, banks)
        # extension, default off = reference semantics: contrast against the banks of ALL ranks
        # (RCCL all-gather of the sampled embeddings; gradients stay rank-local)
        self.global_negatives = bool(config.get('global_negatives', False))
        # similarity-product arithmetic: 'f32' (exact fp32 MFMA) or 'f16x3' (split-f16 MFMA, fp32-equivalent);
        # config key 'mfma_mode', overridable through the debug configuration (mscs_amd/debug.py: DCL_MFMA)
        from ..debug import cfg as _dbg
        self.mfma_mode = _dbg.mfma_mode or config.get('mfma_mode', 'f16x3')
        assert self.mfma_mode in ('f32', 'f16x3'), f"mfma_mode must be 'f32' or 'f16x3', got {self.mfma_mode}"
        if self.label_scaling_mode == 'nn':
            assert self.dominant_mode == 'all', \
                'cannot use label_scaling_mode: "{}" with dominant_mode: "{}" - only "all" is allowed'.format(
                    self.label_scaling_mode, self.dominant_mode)

    def engine_config(self, **over):
        cfg = dict(num_all_classes=int(self.num_all_classes), temperature=float(self.temperature),
                   min_views_per_class=int(self.min_views_per_class),
                   max_views_per_class=int(self.max_views_per_class),
                   max_features_total=int(self.max_features_total),
                   global_negatives=bool(self.global_negatives), mfma=self.mfma_mode)
        cfg.update(over)
        return EngineConfig(**cfg)

    def _note_plan(self, plan, scale):
        """Mirror the logging side effects of _select_views_per_class (ref :64-84)."""
        self._scale = scale
        if plan.log_this_step:
            self.log_this_step = True
            printlog(f'capping views: T={plan.T} V={plan.V} (max_views_per_class='
                     f'{self.max_views_per_class}, max_features_total={self.max_features_total})')

    def forward(self, label: torch.Tensor, features: torch.Tensor):
        if self.cross_scale_contrast and hasattr(features, 'materialize'):
            # (a LazyProjection from a model run with graph key lazy_projector: the 4-tuple below hands out the raw
            # [T, C, V] bank gathered from the MAP, so the map is formed here)
            features = features.materialize()
        terms, st = dense_contrast_terms(self.engine_config(weights=(1.0,)), label, [features])
        self.last_state = st
        sc = st.scales[0]
        self._note_plan(sc.plan, int(label.shape[-1] // features.shape[-1]))
        loss = terms[0]
        if self.cross_scale_contrast:
            f32 = features if features.dtype == torch.float32 else features.float()
            sampled_features = _RawBank.apply(f32, sc)
            sampled_labels = sc.pair_k.to(torch.float32)
            return loss, sampled_features, sampled_labels, False
        return loss
