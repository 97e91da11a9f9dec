"""Training runtime around the hot path, with the reference's call surface
(managers/BaseManager.py:37-555, LoggingManager.py): config dict in, ``load_model`` / ``load_loss`` /
``load_data`` / ``load_optimiser`` / ``train`` / ``train_one_epoch`` / ``validate`` /
``forward_step``; model and loss classes are resolved by NAME from ``mscs_amd.models`` /
``mscs_amd.losses`` exactly like the reference's ``globals()`` lookups (BaseManager.py:439, 474).

Process model (MI355X-first): one process per GPU.  Either the manager spawns them itself
(``parallel: true`` + ``gpu_device: [..]``, reference behaviour, mp.spawn + env:// rendezvous on
127.0.0.1), or it is started under ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) and simply joins the group.  Gradients are all-reduced by DDP over RCCL/xGMI with
``gradient_as_bucket_view=True``; BatchNorm becomes SyncBatchNorm when ``graph.sync_bn``.  The
contrastive loss is rank-local, as in the reference (its ``concat_all_gather`` has no call site).
"""
import datetime
import os
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .. import losses as _losses
from .. import models as _models
from ..datasets import SyntheticSegmentation
from ..losses import LossWrapper
from ..losses.engine import StreamKTimeout
from ..utils import DATASETS_INFO, printlog
from ..utils.config import parse_config
from ..utils.lr_functions import LRFcts
from ..utils.metrics import (out_of_range, take_out_of_range, t_get_confusion_matrix, t_get_mean_iou, t_get_pixel_accuracy,
                             t_metrics_from_confusion_matrix)


def set_seeds(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class BaseManager:
    def __init__(self, configuration, autostart=True):
        self.config = parse_config(configuration)
        cfg = self.config
        self.debugging = cfg['debugging']
        self.dataset = cfg['data']['dataset']
        self.experiment = cfg['data']['experiment']
        self.batch_size = cfg['data']['batch_size']
        self.valid_batch_size = cfg.get('valid_batch_size', 1)
        self.empty_cache = cfg.get('empty_cache', False)
        self.parallel = bool(cfg['parallel'])
        self.allocated_devices = list(cfg['gpu_device'])
        self.n_gpus = len(self.allocated_devices) if self.parallel else 1
        self.world_size = self.n_gpus
        self.rank = 0
        self.epoch = 0
        self.global_step = 0
        self.start_epoch = 0
        self.best_miou = -1.0
        self.best_loss = 1e10
        self._resumed = False
        self.load_report = None
        self.metrics = {}
        self.model = self.loss = self.optimiser = self.scheduler = None
        self.data_loaders, self.samplers = {}, {}
        self.return_features = False
        self.log_dir = cfg.get('log_dir')
        self.under_torchrun = int(os.environ.get('WORLD_SIZE', '1')) > 1
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')       # LoggingManager.py:136-137
        os.environ.setdefault('MASTER_PORT', '29500')
        if autostart:
            self.start()

    def start(self):
        """Reference behaviour of the constructor: build everything and, in parallel training mode,
        run the training loop (BaseManager.py:37-104)."""
        cfg = self.config
        if self.under_torchrun:
            self.setup()
            if cfg['mode'] == 'training':
                self.train()
        elif self.parallel and cfg['mode'] == 'training':
            set_seeds(cfg['seed'])
            mp.spawn(self.distributed_train_worker, nprocs=self.n_gpus, args=())
        else:
            self.setup()

    def setup(self):
        """Build model / loss / data / optimiser for THIS process (joins the process group when
        started under torch.distributed.run); does not train."""
        cfg = self.config
        if self.under_torchrun:
            self.parallel = True
            self.world_size = self.n_gpus = int(os.environ['WORLD_SIZE'])
            self._worker_setup(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ['RANK']),
                               split_batch=cfg.get('batch_is_global', True))
            return
        set_seeds(cfg['seed'])
        use_cuda = cfg['cuda'] and torch.cuda.is_available()
        self.device = torch.device(f'cuda:{self.allocated_devices[0]}' if use_cuda else 'cpu')
        if use_cuda:
            torch.cuda.set_device(self.device)
        self.load_model()
        self.load_loss()
        self.load_data()
        self.load_optimiser()
        self._maybe_load_checkpoint()

    # ------------------------------------------------------------------ processes
    def distributed_train_worker(self, gpu):
        """Entry point of a spawned rank (reference: BaseManager.py:106-145)."""
        self._worker_setup(self.allocated_devices[gpu], gpu)
        self.train()

    def _worker_setup(self, device_index, rank, split_batch=True):
        set_seeds(self.config['seed'])
        self.rank = rank
        use_cuda = self.config['cuda'] and torch.cuda.is_available()
        if use_cuda:
            self.device = torch.device(f'cuda:{device_index}')
            torch.cuda.set_device(self.device)
        else:                                   # CPU ranks (gloo) -- used by the multi-process tests
            self.device = torch.device('cpu')
        if split_batch:
            self.batch_size = int(self.batch_size) // self.n_gpus    # global batch -> per rank (:128)
        if not dist.is_initialized():
            backend = self.config.get('dist_backend', 'nccl' if use_cuda else 'gloo')   # 'nccl' == RCCL on ROCm
            dist.init_process_group(backend=backend, world_size=self.world_size, rank=self.rank)
        self.load_model()
        self.load_loss()
        self.load_data()
        self.load_optimiser()
        self._maybe_load_checkpoint()

    # ------------------------------------------------------------------ construction
    def load_model(self):
        graph = self.config['graph']
        if self.config.get('mode') == 'training' and self.config.get('loss', {}).get('name') == 'LossWrapper' \
                and self.config.get('cuda', False):
            # This manager owns BOTH ends of the model's training outputs -- they go to this package's LossWrapper and
            # metrics, nowhere else (forward_step) -- so a reference JSON config, which knows neither key, gets the fused
            # consumers: the logits stay at 1/4 resolution (up-sampling + cross-entropy / arg-max in one kernel, models/ops.py
            # UpsampledLogits) and the projection heads' last 1x1 convolution is evaluated on the sampled pixels only
            # (models/Projector.LazyProjection).  Same losses and gradients (tests/test_hip_parity.py); a config that says
            # `false` -- or code that calls the model classes directly -- gets the reference's tensors.
            defaulted = [k for k in ('lazy_logits', 'lazy_projector') if k not in graph]
            graph.setdefault('lazy_logits', True)
            graph.setdefault('lazy_projector', True)
            if defaulted:
                # (said out loud, so that a subclass / hook that reads ret['output'] or ret['feats'] as plain tensors
                # knows why it sees UpsampledLogits / LazyProjection objects and which key turns them off)
                printlog(f"[mscs_amd] graph.{' / graph.'.join(defaulted)} not in the config: defaulting to true under this manager -- "
                         "the model's training outputs are UpsampledLogits (logits at 1/4 resolution) and LazyProjection "
                         "(projector rows on the sampled pixels) objects, consumed by this package's LossWrapper and metrics; "
                         "write `false` for the reference's tensors (INTEGRATION.md, 'Runtime behaviour')")
        model_class = getattr(_models, graph['model'])
        self.model = model_class(config=graph, experiment=self.experiment)
        self.return_features = getattr(self.model, 'projector_model', None) is not None \
            and self.config['mode'] == 'training'
        self.model = self.model.to(self.device)
        if self.config.get('channels_last', False):
            self.model = self.model.to(memory_format=torch.channels_last)
        if self.parallel:
            if graph.get('sync_bn', False):
                if self.device.type == 'cuda':
                    from ..models.fused_bn import convert_sync_batchnorm
                    self.model = convert_sync_batchnorm(self.model)
                else:
                    printlog('sync_bn requested but ranks are on CPU: SyncBatchNorm needs GPU modules, '
                             'keeping per-rank BatchNorm statistics')
            self.model = torch.nn.parallel.DistributedDataParallel(
                self.model, device_ids=[self.device] if self.device.type == 'cuda' else None,
                gradient_as_bucket_view=True)
        n_params = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        printlog(f"Using model '{graph['model']}' with backbone '{graph.get('backbone')}' : "
                 f"trainable parameters {n_params}")

    def _bare_model(self):
        return self.model.module if hasattr(self.model, 'module') else self.model

    def load_loss(self):
        lcfg = self.config['loss']
        lcfg['experiment'] = self.experiment
        lcfg['device'] = str(self.device)
        if lcfg['name'] == 'CrossEntropyLoss':
            names = DATASETS_INFO[self.dataset].CLASS_INFO[self.experiment][1]
            self.loss = torch.nn.CrossEntropyLoss(ignore_index=len(names) - 1 if 255 in names else -100)
        else:
            self.loss = getattr(_losses, lcfg['name'])(lcfg)
        self.loss = self.loss.to(self.device)
        if isinstance(self.loss, LossWrapper) and any('DenseContrastive' in t for t in self.loss.loss_classes):
            assert getattr(self._bare_model(), 'projector_model', None) is not None, \
                'model must have projector if DC loss is used'

    def load_data(self):
        dcfg = self.config['data']
        if not dcfg.get('synthetic', False):
            raise NotImplementedError(
                "only data.synthetic=true is available: real Cityscapes / ADE20K loaders "
                "(reference datasets/*.py) are outside the accelerated hot path (SURVEY.md row 15)")
        names = DATASETS_INFO[self.dataset].CLASS_INFO[self.experiment][1]
        size = dcfg.get('transform_values', {}).get('crop_shape', [512, 1024])
        train = SyntheticSegmentation(dcfg.get('synthetic_length', 64), size, len(names),
                                      mode=dcfg.get('synthetic_mode', 'iid'), seed=self.config['seed'])
        valid = SyntheticSegmentation(dcfg.get('synthetic_valid_length', 4), size, len(names),
                                      mode=dcfg.get('synthetic_mode', 'iid'), seed=self.config['seed'] + 1)
        sampler = DistributedSampler(train, num_replicas=self.world_size, rank=self.rank) if self.parallel else None
        self.samplers['train_loader'] = sampler
        self.data_loaders['train_loader'] = DataLoader(
            train, batch_size=self.batch_size, shuffle=sampler is None, sampler=sampler,
            num_workers=dcfg['num_workers'], drop_last=True, pin_memory=self.device.type == 'cuda')
        self.data_loaders['valid_loader'] = DataLoader(valid, batch_size=self.valid_batch_size, shuffle=False,
                                                       num_workers=0)
        self.train_schedule = {e: 'train_loader' for e in range(self.config['train']['epochs'])}

    def _param_groups(self):
        """``train.opt_keys`` {substring: {lr_mult, wd_mult}} -> optimizer parameter groups (reference:
        utils/optimizer_utils.py:34-80): a parameter joins the group of the FIRST key its name contains, every other
        parameter the base group; frozen parameters stay out."""
        tcfg = self.config['train']
        keys = tcfg.get('opt_keys')
        if not keys:
            return self.model.parameters()
        # like the reference (optimizer_utils.py:38): opt_keys without train.weight_decay is a KeyError, not a silent
        # run with zero weight decay (every group carries an explicit value, so the optimiser's default never applies)
        base_lr, base_wd = tcfg['learning_rate'], tcfg['weight_decay']
        groups = {}
        for name, p in self.model.named_parameters():
            if not p.requires_grad:
                continue
            gname, lr, wd = 'base_lr_wd', base_lr, base_wd
            for key, mult in keys.items():
                if key in name:
                    lm, wm = mult.get('lr_mult', 1.0), mult.get('wd_mult', 1.0)
                    gname, lr, wd = f'{key}_lrm{lm}_wdm{wm}', lm * base_lr, wm * base_wd
                    break
            g = groups.setdefault(gname, {'params': [], 'group_name': gname, 'param_names': [], 'lr': lr,
                                          'weight_decay': wd})
            g['params'].append(p)
            g['param_names'].append(name)
        return list(groups.values())

    def load_optimiser(self):
        tcfg = self.config['train']
        params = self._param_groups()
        optim = tcfg.get('optim', 'Adam')
        # single-pass ("fused") parameter updates where torch has them for the device: the same update rule in ONE kernel pass
        # per tensor list instead of the foreach form's three (weight decay, momentum, step): 1.4 -> 0.5 ms of the W48 step,
        # at its very end where nothing overlaps (train.fused_optimizer = false keeps the foreach form)
        fused = {}
        params = list(params)
        flat = [p for g in params for p in (g['params'] if isinstance(g, dict) else [g])]
        if tcfg.get('fused_optimizer', True) and self.config.get('cuda', False) and \
                all(p.is_cuda and p.dtype == torch.float32 for p in flat):
            fused = {'fused': True}
        if optim == 'SGD':
            self.optimiser = torch.optim.SGD(params, lr=tcfg['learning_rate'], momentum=tcfg.get('momentum', 0.9),
                                             weight_decay=tcfg.get('weight_decay', 0.0005), **fused)
        elif optim == 'Adam':
            self.optimiser = torch.optim.Adam(params, lr=tcfg['learning_rate'])
        elif optim == 'AdamW':
            self.optimiser = torch.optim.AdamW(params, lr=tcfg['learning_rate'],
                                               betas=tuple(tcfg.get('betas', (0.9, 0.999))),
                                               weight_decay=tcfg.get('weight_decay', 0.01), **fused)
        else:
            raise ValueError(f"optimizer {optim} not recognized")
        # an invalid gradient of the contrastive loss (stream-K backward: a timed-out hand-over, losses/engine.py) must never
        # reach the parameters: checked right before every update (the error word was copied to the host at the START of the
        # backward pass, so this normally does not wait)
        self.optimiser.register_step_pre_hook(lambda *_a, **_k: self._check_backward())
        if tcfg['lr_batchwise']:
            total = sum(len(self.data_loaders[self.train_schedule[e]]) for e in range(tcfg['epochs']))
        else:
            total = tcfg['epochs']
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.optimiser, lr_lambda=LRFcts(tcfg, list(tcfg.get('lr_restarts', [])), max(total, 2)))

    # ------------------------------------------------------------------ loops
    def forward_step(self, img, lbl, **kwargs):
        raise NotImplementedError

    def _check_backward(self):
        if self.device.type == 'cuda':
            from ..losses.engine import streamk_check
            streamk_check()

    def train(self):
        tcfg = self.config['train']
        for self.epoch in range(self.start_epoch, tcfg['epochs']):
            sampler = self.samplers.get(self.train_schedule[self.epoch])
            if sampler is not None:
                sampler.set_epoch(self.epoch)
            self.train_one_epoch()
            if (self.epoch + 1) % self.config.get('valid_freq', 1) == 0 or self.epoch == tcfg['epochs'] - 1:
                self.validate()
        if self.parallel and dist.is_initialized():
            dist.barrier()

    def _upload(self, img, lbl):
        """H2D of a batch on a dedicated input stream (+ the label's int64 conversion), and the event that marks
        it complete: the loss's label stage waits for THAT event only (engine.stage_labels), so it -- and the host
        plan built from it -- can run while the GPU is still in the previous step's backward."""
        if self.device.type != 'cuda':
            return img.to(self.device), lbl.to(self.device), None
        if getattr(self, '_in_stream', None) is None:
            self._in_stream = torch.cuda.Stream(device=self.device, priority=-1)    # own hardware queue
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._in_stream):
            img = img.to(self.device, non_blocking=True)
            lbl = lbl.to(self.device, non_blocking=True).long()
            ready = torch.cuda.Event()
            ready.record(self._in_stream)
        cur.wait_event(ready)
        img.record_stream(cur)
        lbl.record_stream(cur)
        return img, lbl, ready

    def train_one_epoch(self):
        self.model.train()
        t_prev = time.perf_counter()
        for batch_num, batch in enumerate(self.data_loaders[self.train_schedule[self.epoch]]):
            img, lbl, ready = self._upload(batch[0], batch[1])
            self.optimiser.zero_grad()
            ret = self.forward_step(img, lbl, label_ready=ready)
            ret['loss'].backward()
            try:
                self.optimiser.step()
            except StreamKTimeout as e:
                # raised by the pre-step hook BEFORE anything was applied (on every rank: the error word is all-reduced).  The
                # library has switched to the column-split backward: repeat the step on the same batch.  (The repeated forward
                # updates the norms' running statistics and draws the sampling permutations a second time.)
                printlog(f'[warning] {e}\n          repeating step {self.global_step}')
                self.optimiser.zero_grad()
                ret = self.forward_step(img, lbl, label_ready=ready)
                ret['loss'].backward()
                self.optimiser.step()
            if self.scheduler is not None and self.config['train']['lr_batchwise']:
                self.scheduler.step()
            if batch_num == 2 and self.debugging:
                break
            self.step_metrics(batch_num, ret, lbl, (time.perf_counter() - t_prev) * 1e3)
            t_prev = time.perf_counter()
            self.global_step += 1
        self.flush_logging()
        if self.scheduler is not None and not self.config['train']['lr_batchwise']:
            self.scheduler.step()

    def step_metrics(self, batch_num, ret, lbl, ms):
        """The per-step tail of the reference's loop (HRNet_Manager.py:117-121): confusion matrix (one HIP pass over
        the logits, utils/metrics.py), pixel accuracies, mIoU, logging."""
        cm = t_get_confusion_matrix(ret['output'], lbl, self.dataset)
        pa, pac, miou = t_metrics_from_confusion_matrix(cm)
        self.train_logging(batch_num, ret['loss'], pa, pac, miou, ms)

    def train_logging(self, batch_num, loss, pa, pac, miou, ms):
        """All scalars of the step travel in ONE asynchronous D2H (the reference issues one blocking .item() per
        logged value) and are read one step LATER, when the copy has long finished: the host never waits for the
        device here.  ``self.metrics`` therefore describes the previous step until ``flush_logging()``."""
        keys, vals = ['loss', 'pa', 'pac', 'miou'], [loss.detach().float(), pa, pac, miou]
        if isinstance(self.loss, LossWrapper):
            for k, v in self.loss.loss_vals.items():
                if torch.is_tensor(v):
                    keys.append(k)
                    vals.append(v.float())
        if loss.is_cuda:          # targets outside the class range, counted by the confusion-matrix kernel
            keys.append('_oob')
            vals.append(take_out_of_range(loss.device))           # this step's count; the counter restarts at 0
        dev_vals = torch.stack([v.reshape(()) for v in vals])
        self.flush_logging()                                        # the PREVIOUS step's record
        if loss.is_cuda:
            host = getattr(self, '_log_host', None)
            if host is None or host.numel() < dev_vals.numel():
                host = self._log_host = torch.empty(max(64, dev_vals.numel()), dtype=torch.float32, pin_memory=True)
            host[:dev_vals.numel()].copy_(dev_vals, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._log_pending = (keys, host, dev_vals.numel(), ev, batch_num, ms)
        else:
            self._log_pending = (keys, dev_vals, dev_vals.numel(), None, batch_num, ms)
            self.flush_logging()

    def flush_logging(self):
        pend, self._log_pending = getattr(self, '_log_pending', None), None
        if pend is None:
            return
        keys, host, n, ev, batch_num, ms = pend
        if ev is not None:
            ev.synchronize()
        vals = host[:n].tolist()
        if keys[-1] == '_oob':
            keys = keys[:-1]
            if vals.pop() > 0:
                raise RuntimeError('Class values must be smaller than num_classes.')   # F.one_hot's error in the
                # reference's t_get_confusion_matrix (utils/torch_utils.py:172-177), reported one step late
        self.metrics.update(zip(keys, vals))
        if self.rank == 0 and (batch_num % self.config.get('log_every_n_steps', 10) == 0):
            terms = ' '.join(f'{k}:{v:.4f}' for k, v in zip(keys, vals))
            printlog(f'ep {self.epoch} it {batch_num} t:{ms:.0f}ms {terms}')

    @torch.no_grad()
    def validate(self):
        if self.rank != 0:          # rank 0 only, like the reference (HRNet_Manager.py:150-156)
            return None
        self.model.eval()
        cm = 0
        for i, batch in enumerate(self.data_loaders['valid_loader']):
            img, lbl = batch[0].to(self.device), batch[1].to(self.device)
            out = self._bare_model()(img.float())
            logits = out[0] if isinstance(out, (tuple, list)) and out[0].dim() == 4 and len(out) == 2 else out
            if isinstance(logits, (tuple, list)):
                logits = logits[1]
            cm = cm + t_get_confusion_matrix(logits, lbl, self.dataset)
            if i + 1 >= self.config.get('max_valid_imgs', 10):
                break
        miou = float(t_get_mean_iou(cm).item())
        if torch.is_tensor(cm) and cm.is_cuda and int(take_out_of_range(cm.device).item()) > 0:
            # a validation label outside the class range: F.one_hot's error in the reference's confusion matrix
            # (utils/torch_utils.py:172-177), raised here instead of by the next training epoch's first flush
            self.model.train()
            raise RuntimeError('Class values must be smaller than num_classes.')
        self.metrics['final_miou'] = miou
        self.metrics['final_miou_step'] = self.global_step - 1
        is_best = miou > self.best_miou
        self.best_miou = max(self.best_miou, miou)
        printlog(f'validation epoch {self.epoch}: mIoU {miou:.4f}')
        # checkpoints as in valid_logging (LoggingManager.py:280-284): the best so far, and one every
        # log_every_n_epochs plus the final epoch; `save_checkpoints: false` turns them off (benchmarks)
        if self.config['mode'] == 'training' and self.config.get('save_checkpoints', True) \
                and self.optimiser is not None:
            if is_best:
                self.save_checkpoint(is_best=True)
            every = self.config.get('log_every_n_epochs', 100)
            if (self.epoch % every == 0 and self.epoch > 0) or self.epoch == self.config['train']['epochs'] - 1:
                self.save_checkpoint(is_best=False)
        self.model.train()
        return miou

    # ------------------------------------------------------------------ checkpoints
    def _log_dir(self):
        """``<log_path>/<run_id>`` like the reference (LoggingManager.py:81-91); created on first use so that runs
        which never save (benchmarks, tests) leave nothing behind."""
        if self.log_dir is None:
            run_id = self.config.get('run_id')
            if run_id is None:
                run_id = '{:%Y%m%d_%H%M%S}_e{}'.format(datetime.datetime.now(), self.experiment)
                if 'name' in self.config:
                    run_id = '__'.join((run_id, str(self.config['name'])))
            self.log_dir = os.path.join(self.config.get('log_path', 'logging'), run_id)
        os.makedirs(os.path.join(str(self.log_dir), 'chkpts'), exist_ok=True)
        return str(self.log_dir)

    def checkpoint_state(self, is_best=False):
        """The reference's checkpoint dictionary, key for key (LoggingManager.py:293-313)."""
        state = {'global_step': self.global_step - 1,       # the reference saves after train_logging's increment
                 'epoch': self.epoch,                        # absolute: train() resumes at start_epoch
                 'model_state_dict': self.model.state_dict(),
                 'optimiser_state_dict': self.optimiser.state_dict(),
                 'best_loss': self.best_loss,
                 'best_miou': self.best_miou,
                 'final_miou': self.metrics.get('final_miou', 0),
                 'final_miou_step': self.metrics.get('final_miou_step', 0),
                 'is_best': bool(is_best)}
        if self.scheduler is not None:
            state['scheduler_state_dict'] = self.scheduler.state_dict()
        return state

    def save_checkpoint(self, is_best=False, path=None):
        """``chkpts/chkpt_best.pt`` or ``chkpts/chkpt_epoch_NNN.pt`` under the run's log directory (or an explicit
        ``path``), written by rank 0 only; returns the path."""
        if self.rank != 0:
            return None
        state = self.checkpoint_state(is_best)
        if path is None:
            name = 'chkpt_best.pt' if is_best else 'chkpt_epoch_{:03d}.pt'.format(state['epoch'])
            path = os.path.join(self._log_dir(), 'chkpts', name)
        torch.save(state, path)
        printlog(f'Checkpoint saved: {path}')
        return path

    def _resolve_checkpoint(self, spec, chkpt_type):
        """``spec``: a .pt file, a run directory (with ``chkpts/``), or a run id under ``log_path``; picks
        ``chkpt_best.pt`` or the last ``chkpt_epoch_*.pt`` like LoggingManager.py:327-344."""
        spec = str(spec)
        if os.path.isfile(spec):
            return spec
        base = spec if os.path.isdir(spec) else os.path.join(self.config.get('log_path', 'logging'), spec)
        ck = os.path.join(base, 'chkpts') if os.path.isdir(os.path.join(base, 'chkpts')) else base
        if not os.path.isdir(ck):
            raise FileNotFoundError(f"load_checkpoint: '{spec}' is neither a checkpoint file nor a run directory")
        names = sorted(os.listdir(ck))
        epochs = [n for n in names if n.startswith('chkpt_epoch_')]
        if chkpt_type == 'last':
            if not epochs:
                raise ValueError("No checkpoint of type 'last' found.")
            return os.path.join(ck, epochs[-1])
        if 'chkpt_best.pt' in names:
            return os.path.join(ck, 'chkpt_best.pt')
        if epochs:
            printlog("No checkpoint of type 'best' found; loading the last one.")
            return os.path.join(ck, epochs[-1])
        raise ValueError(f'Neither chkpt of type "best" nor of type "last" was found in {names}')

    @staticmethod
    def match_module_prefix(state, target_keys):
        """Rename ``state``'s keys so that DDP's ``module.`` prefix agrees with the model it is loaded into, in
        BOTH directions (a single-GPU save into a DDP-wrapped model and vice versa; the reference only strips,
        LoggingManager.py:353-354 / check_module_prefix)."""
        want = any(k.startswith('module.') for k in target_keys)
        out = {}
        for k, v in state.items():
            has = k.startswith('module.')
            if want and not has:
                k = 'module.' + k
            elif has and not want:
                k = k[len('module.'):]
            out[k] = v
        return out

    def load_checkpoint(self, spec, chkpt_type='best', strict_keys=True):
        """Load a checkpoint in the reference's dictionary layout (LoggingManager.py:321-368).  Missing and
        unexpected keys are logged; a checkpoint that matches NO parameter of the model raises (the silent
        ``strict=False`` no-op of a prefix mismatch)."""
        path = self._resolve_checkpoint(spec, chkpt_type)
        chk = torch.load(path, map_location=self.device, weights_only=False)
        target = self.model.state_dict()
        state = self.match_module_prefix(chk['model_state_dict'], target.keys())
        matched = [k for k in state if k in target]
        if not matched:
            raise RuntimeError(f'checkpoint {path} shares no parameter name with the model '
                               f'(first keys: {list(state)[:3]} vs {list(target)[:3]})')
        ret = self.model.load_state_dict(state, strict=False)
        if ret.missing_keys or ret.unexpected_keys:
            printlog(f'load_state_dict: {len(matched)} tensors loaded, missing {len(ret.missing_keys)} '
                     f'{ret.missing_keys[:5]}, unexpected {len(ret.unexpected_keys)} {ret.unexpected_keys[:5]}')
            if strict_keys and self.config.get('strict_checkpoint', False):
                raise RuntimeError('checkpoint does not match the model (strict_checkpoint)')
        self.load_report = ret
        if self.config['mode'] == 'training':
            if 'optimiser_state_dict' in chk and self.optimiser is not None:
                self.optimiser.load_state_dict(chk['optimiser_state_dict'])
            if chk.get('scheduler_state_dict') and self.scheduler is not None:
                self.scheduler.load_state_dict(chk['scheduler_state_dict'])
            self.start_epoch = chk.get('epoch', -1) + 1
            self.global_step = chk.get('global_step', 0)
            self.best_loss = chk.get('best_loss', 1e10)
            self.best_miou = chk.get('best_miou', -1.0)
            self.metrics['final_miou'] = chk.get('final_miou')
            self._resumed = True
        printlog(f'rank {self.rank}: checkpoint loaded: {path} (type {chkpt_type})')
        return path

    def _maybe_load_checkpoint(self):
        """``load_checkpoint`` / ``load_last`` config keys (BaseManager.py:76-82, 139-144)."""
        if 'load_checkpoint' in self.config:
            kind = 'last' if self.config.get('load_last', False) else 'best'
            self.load_checkpoint(self.config['load_checkpoint'], kind)
