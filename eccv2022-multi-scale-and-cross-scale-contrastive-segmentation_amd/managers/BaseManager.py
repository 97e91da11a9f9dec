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
from ..utils import DATASETS_INFO, printlog
from ..utils.config import parse_config
from ..utils.lr_functions import LRFcts
from ..utils.metrics import t_get_confusion_matrix, t_get_mean_iou, t_get_pixel_accuracy


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

    # ------------------------------------------------------------------ construction
    def load_model(self):
        graph = self.config['graph']
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

    def load_optimiser(self):
        tcfg = self.config['train']
        params = self.model.parameters()
        optim = tcfg.get('optim', 'Adam')
        if optim == 'SGD':
            self.optimiser = torch.optim.SGD(params, lr=tcfg['learning_rate'], momentum=tcfg.get('momentum', 0.9),
                                             weight_decay=tcfg.get('weight_decay', 0.0005))
        elif optim == 'Adam':
            self.optimiser = torch.optim.Adam(params, lr=tcfg['learning_rate'])
        elif optim == 'AdamW':
            self.optimiser = torch.optim.AdamW(params, lr=tcfg['learning_rate'],
                                               betas=tuple(tcfg.get('betas', (0.9, 0.999))),
                                               weight_decay=tcfg.get('weight_decay', 0.01))
        else:
            raise ValueError(f"optimizer {optim} not recognized")
        if tcfg['lr_batchwise']:
            total = sum(len(self.data_loaders[self.train_schedule[e]]) for e in range(tcfg['epochs']))
        else:
            total = tcfg['epochs']
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.optimiser, lr_lambda=LRFcts(tcfg, list(tcfg.get('lr_restarts', [])), max(total, 2)))

    # ------------------------------------------------------------------ loops
    def forward_step(self, img, lbl, **kwargs):
        raise NotImplementedError

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

    def train_one_epoch(self):
        self.model.train()
        t_prev = time.perf_counter()
        for batch_num, batch in enumerate(self.data_loaders[self.train_schedule[self.epoch]]):
            img, lbl = batch[0], batch[1]
            img = img.to(self.device, non_blocking=True)
            lbl = lbl.to(self.device, non_blocking=True)
            self.optimiser.zero_grad()
            ret = self.forward_step(img, lbl)
            ret['loss'].backward()
            self.optimiser.step()
            if self.scheduler is not None and self.config['train']['lr_batchwise']:
                self.scheduler.step()
            if batch_num == 2 and self.debugging:
                break
            cm = t_get_confusion_matrix(ret['output'], lbl, self.dataset)
            pa, pac = t_get_pixel_accuracy(cm)
            miou = t_get_mean_iou(cm)
            now = time.perf_counter()
            self.train_logging(batch_num, ret['loss'], pa, pac, miou, (now - t_prev) * 1e3)
            t_prev = now
            self.global_step += 1
        if self.scheduler is not None and not self.config['train']['lr_batchwise']:
            self.scheduler.step()

    def train_logging(self, batch_num, loss, pa, pac, miou, ms):
        """One D2H for all scalars of the step (the reference issues one .item() per logged value)."""
        keys, vals = ['loss', 'pa', 'pac', 'miou'], [loss.detach().float(), pa, pac, miou]
        if isinstance(self.loss, LossWrapper):
            for k, v in self.loss.loss_vals.items():
                if torch.is_tensor(v):
                    keys.append(k)
                    vals.append(v.float())
        host = torch.stack([v.reshape(()) for v in vals]).cpu().tolist()
        self.metrics = dict(zip(keys, host))
        if self.rank == 0 and (batch_num % self.config.get('log_every_n_steps', 10) == 0):
            terms = ' '.join(f'{k}:{v:.4f}' for k, v in self.metrics.items())
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
        self.best_miou = max(self.best_miou, miou)
        printlog(f'validation epoch {self.epoch}: mIoU {miou:.4f}')
        self.model.train()
        return miou

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, path):
        """Same dictionary layout as the reference (LoggingManager.py:293-319)."""
        torch.save({'global_step': self.global_step, 'epoch': self.epoch,
                    'model_state_dict': self.model.state_dict(),
                    'optimiser_state_dict': self.optimiser.state_dict(),
                    'scheduler_state_dict': self.scheduler.state_dict() if self.scheduler else None,
                    'best_miou': self.best_miou}, path)

    def load_checkpoint(self, path):
        """Loads a reference-format checkpoint; strips DDP's ``module.`` prefix when not parallel."""
        chk = torch.load(path, map_location=self.device)
        state = chk['model_state_dict']
        if not self.parallel:
            state = {(k[7:] if k.startswith('module.') else k): v for k, v in state.items()}
        self.model.load_state_dict(state, strict=False)
        if 'optimiser_state_dict' in chk and self.optimiser is not None:
            self.optimiser.load_state_dict(chk['optimiser_state_dict'])
        if chk.get('scheduler_state_dict') and self.scheduler is not None:
            self.scheduler.load_state_dict(chk['scheduler_state_dict'])
        self.start_epoch = chk.get('epoch', -1) + 1
        self.global_step = chk.get('global_step', 0)
        self.best_miou = chk.get('best_miou', -1.0)
