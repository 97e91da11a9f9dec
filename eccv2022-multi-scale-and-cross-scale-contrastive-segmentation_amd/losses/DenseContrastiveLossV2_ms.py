"""Multi-scale + cross-scale dense contrastive loss -- drop-in for the reference class of the same
name (losses/DenseContrastiveLossV2_ms.py:12-161).  All scales and cross-scale terms of one step
run as ONE autograd node over libdcl_hip.so (one host sync for the label histograms)."""
import torch
import torch.nn as nn

from ..utils import DATASETS_INFO, is_distributed, printlog
from .DenseContrastiveLossV2 import DenseContrastiveLossV2 as DCV2
from .engine import PreSampleFailed, dense_contrast_terms, presample
from .plan import NoQualifyingPair


class DenseContrastiveLossV2_ms(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.parallel = is_distributed()
        self.experiment = config['experiment']
        self.dataset = config['dataset']
        names = DATASETS_INFO[self.dataset].CLASS_INFO[self.experiment][1]
        self.num_all_classes = len(names)
        self.num_real_classes = self.num_all_classes - 1 if 255 in names else self.num_all_classes
        self.ignore_class = (len(names) - 1) if 255 in names else -1
        self.scales = config['scales'] if 'scales' in config else 2
        self.weights = config['weights'] if 'weights' in config else [1.0] * self.scales
        assert self.scales == len(self.weights), \
            f'given dc loss number of scales [{self.scales}] not equal len of weights {self.weights}'
        self.losses = []
        self.eps = torch.tensor(1e-10)
        self.meta = {name: (0.0, 0.0) for name in names.values()}
        self.cross_scale_contrast = config['cross_scale_contrast'] if 'cross_scale_contrast' in config else False
        # reference quirk (ms:28): the VALUE of 'cross_scale_temperature' is ignored -- its presence
        # selects the constant 0.1, its absence selects config['temperature']
        self.cross_scale_temperature = config['temperature'] if 'cross_scale_temperature' not in config else 0.1
        self.detach_cs_deepest = config['detach_deepest'] if 'detach_deepest' in config else False
        self.w_high_low = config['w_high_low'] if 'w_high_low' in config else 1.0
        self.w_high_mid = config['w_high_mid'] if 'w_high_mid' in config else 1.0
        self.ms_losses = []
        self.cs_losses = []
        self.last_state = None
        self._geoms = {}                  # label shape -> [(scale, h, w)] seen in earlier steps
        self._staged = None
        self._side_stream = None
        self._emulated_peers = None       # test hook: (rank, peer_banks, peer_layouts) of virtual ranks
        printlog(f'defining dcv2 ms loss with number of scales {self.scales} and weights {self.weights}')
        printlog(f'using cross scale contrast {self.cross_scale_contrast}')
        for s in range(self.scales):
            setattr(self, f'DCV2_scale{s}', DCV2(config))
        if self.cross_scale_contrast:
            printlog(f'using cross-scale contrast with detach_cs_deepest set to {self.detach_cs_deepest}, '
                     f'w_high_low: {self.w_high_low}, w_high_mid: {self.w_high_mid}')

    def prepare(self, label: torch.Tensor, ready_event=None):
        """Optional, call BEFORE the model forward is enqueued: runs everything of the loss that depends on the
        labels only -- stride-sample + class histograms of every scale, the 960-byte D2H, the host-side sampling
        plan (the reference's randperm draws, same order) and the pixel selection kernel -- on a side stream, so
        that none of it waits behind the model forward and the forward pass is not followed by ~1.5 ms of host work
        with the GPU idle.  A no-op until one forward has been seen for this label shape (the strides come from the
        feature maps).  It DOES consume the CPU generator: the randperm draws of this step happen here, in the
        reference's order, instead of inside forward() -- so call it only for a step whose forward() will evaluate
        this loss on the same label tensor (LossWrapper.prepare skips components that are not in ``loss_list``).
        A planning error (e.g. no (image, class) pair with ``min_views_per_class`` pixels) is not raised here: it is
        kept and re-raised by forward() from inside the autograd function, i.e. behind the all-rank agreement of the
        shared negative bank (engine.agree_or_raise) -- raising here would leave the other ranks waiting in their
        first collective."""
        geoms = self._geoms.get(tuple(label.shape))
        if geoms is None or not label.is_cuda:
            return False
        if self._side_stream is None:
            # high priority: its own hardware queue, so that the label stage is not queued behind the previous step's
            # backward (HIP maps the ordinary streams of a process onto a few shared hardware queues)
            self._side_stream = torch.cuda.Stream(device=label.device, priority=-1)
        try:
            self._staged = presample(self._engine_cfg(), label, geoms, bool(self.cross_scale_contrast),
                                     self._side_stream, ready_event=ready_event)
        except NoQualifyingPair as e:          # data-dependent: surfaces in forward(), see the docstring.  Anything else
            self._staged = PreSampleFailed(e, label)       # (a shape / ctypes / launch error) is a bug and raises HERE
        return True

    def discard_prepared(self):
        """Drop what prepare() staged (the step turned out not to evaluate this loss).  A parked planning error is logged
        before it goes: its randperm draws are spent, i.e. the run has left the reference's RNG sequence at this step."""
        if isinstance(self._staged, PreSampleFailed):
            printlog(f'DenseContrastiveLossV2_ms: dropping a planning error parked by prepare() for a step that does not '
                     f'evaluate this loss: {self._staged.error}')
        self._staged = None

    def _engine_cfg(self):
        return self.DCV2_scale0.engine_config(weights=tuple(float(w) for w in self.weights),
                                              cross_scale_contrast=bool(self.cross_scale_contrast),
                                              cross_scale_temperature=float(self.cross_scale_temperature),
                                              detach_deepest=bool(self.detach_cs_deepest),
                                              w_high_low=float(self.w_high_low), w_high_mid=float(self.w_high_mid))

    def forward(self, label: torch.Tensor, features: list, **kwargs):
        self.cs_losses = []
        self.ms_losses = []
        S = self.scales
        sub0 = self.DCV2_scale0
        with_cross = bool(self.cross_scale_contrast)
        if with_cross:
            assert S > 1 and len(features) > 1
        cfg = self._engine_cfg()
        feats = [features[s] for s in range(S)]
        staged, self._staged = self._staged, None
        terms, st = dense_contrast_terms(cfg, label, feats, staged=staged,
                                         emulated_peers=self._emulated_peers)
        self.last_state = st
        self._geoms[tuple(label.shape)] = [(int(label.shape[-1] // f.shape[-1]), f.shape[2], f.shape[3])
                                           for f in feats]
        for s in range(S):
            getattr(self, f'DCV2_scale{s}')._note_plan(st.scales[s].plan,
                                                      int(label.shape[-1] // feats[s].shape[-1]))
        wkey = (tuple(t.weight for t in st.terms), terms.device)
        if getattr(self, '_wvec_key', None) != wkey:      # static per config: upload once, not per step
            self._wvec = torch.tensor(wkey[0], dtype=torch.float32, device=terms.device)
            self._wvec_key = wkey
        loss = (terms * self._wvec).sum()
        det = terms.detach()
        self.ms_losses = [det[s] for s in range(S)]
        if with_cross:
            # ms:66-80 -- the high->low term is logged only when the deep bank is NOT detached
            if not self.detach_cs_deepest:
                self.cs_losses.append(det[S])
            if S > 2:
                self.cs_losses.append(det[S + 1])
        return loss
