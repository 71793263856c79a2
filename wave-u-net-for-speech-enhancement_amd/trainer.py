"""train-step driver, selectable as the reference's trainer plugin

    "trainer": {"module": "wave-u-net-for-speech-enhancement_amd.trainer", "main": "Trainer", "epochs": ..., ...}

Same constructor as the reference's Trainer (/root/reference/trainer/trainer.py:13-25, called from train.py:41-49)
and the same `.train()` entry (train.py:51), covering the hot loop only (trainer/trainer.py:27-43,
base_trainer.py:187-197): validation, TensorBoard and metrics are out of scope.
Differences from the reference loop, all on the f2 list:
  * one process per GPU: launched under torchrun (RANK / WORLD_SIZE / LOCAL_RANK in the environment) the constructor joins the
    process group itself (backend "nccl" = RCCL on a GPU box, "gloo" without one), pins the process to its GPU, re-wraps the
    incoming DataLoader with a DistributedSampler (train.py:15-21 builds it without one; the reference's batch_size is the
    GLOBAL batch that nn.DataParallel splits, base_trainer.py:26-27, so each rank loads batch_size / world_size items) and
    attaches the RCCL gradient all-reduce to the model instead of nn.DataParallel;
  * host->device copies are non_blocking from pinned memory (train.py:20 sets pin_memory but copies synchronously);
  * the loss is accumulated on the device and read back once per epoch (trainer.py:40 syncs every step);
  * after a few eager steps the whole step - forward, loss, backward, Adam (optim.FusedAdam with its device-side step counter) - is
    captured once in a hipGraph (torch.cuda.CUDAGraph) on static input buffers and replayed: ~230 kernel launches become one graph
    launch per step (default on one GPU with FusedAdam; `"graph": false` in the trainer config or WUNET_GRAPH=0 keeps eager launches;
    hyper-parameters that change every step - a per-step LR schedule - make it fall back to eager by itself);
  * checkpoints keep the reference's names and keys (base_trainer.py:83-124) so either side can resume; only rank 0 writes.
"""
import os
from pathlib import Path

import torch
import torch.distributed as dist

from .optim import FusedAdam
from .parallel import GradSync

GRAPH_WARMUP_STEPS = 3
# A captured step graph is given up for eager launches when MAX_RECAPTURES graphs IN A ROW did not earn their capture back: a graph
# earns REPLAY_CREDIT_S per replay (what a replay saves over the eager step when the loop synchronises: host launch jitter, 5.27 vs
# 5.5 ms median at the BASELINE size; back to back the two are level) and costs the measured wall time of its capture + instantiate;
# fewer than MIN_REPLAYS replays never count as earned.  A schedule that changes lr / betas / eps every step - or every 3 or 20 steps -
# therefore ends on eager launches after three captures, a per-epoch scheduler over long epochs keeps its graph.
MAX_RECAPTURES = 3
MIN_REPLAYS = 16
REPLAY_CREDIT_S = 0.25e-3


class Trainer:
    def __init__(self, config, resume, model, loss_function, optimizer, train_dataloader, validation_dataloader=None):
        env_world = int(os.environ.get("WORLD_SIZE", "1"))
        self._own_pg = False
        if env_world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
            self._own_pg = True
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            self.device = torch.device("cuda", local_rank)
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device("cpu")
        self.model = model.to(self.device)
        self.loss_function = loss_function
        self.optimizer = optimizer
        fused = isinstance(optimizer, FusedAdam)
        tcfg = config["trainer"]
        # "native_rccl": the bucket all-reduces through the library's own RCCL entry (wunet_comm_*, capturable in the step graph)
        # instead of torch.distributed's process group; default: torch.distributed (measured on hardware only at world size 1 so far)
        native = bool(tcfg.get("native_rccl", os.environ.get("WUNET_NATIVE_RCCL", "0") not in ("", "0"))) and self.device.type == "cuda"
        if self.world > 1:
            from .parallel import NativeComm
            self.model.grad_sync = GradSync(scale_in_optimizer=fused, comm=NativeComm() if native else None)
            if fused:
                optimizer.grad_scale = 1.0 / self.world
        self.train_data_loader = self._shard_loader(train_dataloader)
        self.validation_data_loader = validation_dataloader      # accepted for signature compatibility, unused
        self.epochs = tcfg["epochs"]
        self.save_checkpoint_interval = tcfg.get("save_checkpoint_interval", 0)
        # default ON where it can be used (one GPU, or the native RCCL entry; fused Adam): the replay is as fast as the eager
        # launches back to back and a loop that synchronises every step (trainer/trainer.py:40 `loss.item()`) loses the host's launch
        # jitter (profiles/r4_graph_vs_eager.txt).  "graph": false / WUNET_GRAPH=0 turns it off.
        self.use_graph = bool(tcfg.get("graph", os.environ.get("WUNET_GRAPH", "1") not in ("", "0")))
        if self.use_graph and not (fused and self.device.type == "cuda" and (self.world == 1 or native)):
            # (torch.distributed's collectives stay outside a captured graph: their eager path overlaps them with the backward already;
            #  the native RCCL entry enqueues on the captured streams and is replayed with the step)
            self.use_graph = False
        if self.use_graph:
            optimizer.device_step = True
        self._graph = None
        self._graph_sig = None
        self._static = None
        self._eager_steps = 0
        self._recaptures = 0          # consecutive graphs, dropped for a changed lr / betas / eps, that did not earn their capture back
        self._replays = 0
        self._capture_s = 0.0         # wall time of the live graph's capture
        self.start_epoch = 1
        self.best_score = float("-inf")
        root = Path(os.path.expanduser(config.get("root_dir", "."))).absolute() / config.get("experiment_name", "exp")
        self.checkpoints_dir = root / "checkpoints"
        self.epoch_losses = []
        if resume:
            self._resume_checkpoint()

    # ---- data: every rank its own shard of each global batch (DataParallel's split, base_trainer.py:26-27)
    def _shard_loader(self, loader):
        if self.world == 1 or loader is None:
            return loader
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        if isinstance(getattr(loader, "sampler", None), DistributedSampler):
            return loader
        if not isinstance(loader, DataLoader):
            raise TypeError("Trainer under torchrun needs a torch DataLoader (or one that already carries a DistributedSampler) "
                            "to shard the data; got " + type(loader).__name__)
        if loader.batch_size is None or loader.batch_size % self.world != 0:
            raise ValueError(f"batch_size={loader.batch_size} (the global batch, as nn.DataParallel would split it) must be a "
                             f"multiple of the world size {self.world}")
        shuffle = isinstance(loader.sampler, torch.utils.data.RandomSampler)
        # the reference's loader (train.py:15-21) keeps a ragged last batch: so does every rank here (the sampler pads the
        # dataset to equal per-rank counts instead of dropping items), and the loader's own options travel along
        sampler = DistributedSampler(loader.dataset, num_replicas=self.world, rank=self.rank, shuffle=shuffle, drop_last=False)
        extra = {}
        if loader.num_workers > 0:
            extra = dict(persistent_workers=loader.persistent_workers, prefetch_factor=loader.prefetch_factor)
        return DataLoader(loader.dataset, batch_size=loader.batch_size // self.world, sampler=sampler,
                          num_workers=loader.num_workers, pin_memory=loader.pin_memory, drop_last=loader.drop_last,
                          collate_fn=loader.collate_fn, worker_init_fn=loader.worker_init_fn, generator=loader.generator,
                          timeout=loader.timeout, **extra)

    # ---- checkpoints: reference layout (base_trainer.py:62-124)
    def _resume_checkpoint(self):
        path = self.checkpoints_dir / "latest_model.tar"
        assert path.exists(), f"{path} does not exist, can not load latest checkpoint."
        ckpt = torch.load(path.as_posix(), map_location=self.device)
        self.start_epoch = ckpt["epoch"] + 1
        self.best_score = ckpt["best_score"]
        self.optimizer.load_state_dict(ckpt["optimizer"])
        self.model.load_state_dict(ckpt["model"])

    def _save_checkpoint(self, epoch):
        if self.rank != 0:
            return
        self.checkpoints_dir.mkdir(parents=True, exist_ok=True)
        state = {"epoch": epoch, "best_score": self.best_score, "optimizer": self.optimizer.state_dict(),
                 "model": {k: v.cpu() for k, v in self.model.state_dict().items()}}
        torch.save(state, (self.checkpoints_dir / "latest_model.tar").as_posix())
        torch.save(state["model"], (self.checkpoints_dir / f"model_{str(epoch).zfill(4)}.pth").as_posix())

    # ---- one step of the hot loop (trainer/trainer.py:34-38)
    def _eager_step(self, mixture, clean):
        self.optimizer.zero_grad(set_to_none=True)
        enhanced = self.model(mixture)
        loss = self.loss_function(clean, enhanced)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _capture(self, mixture, clean):
        """The step on static buffers, captured once (after the eager warm-up steps have created every lazily-built object:
        contexts, side stream, LDS attributes, optimiser state, device step counter)."""
        self._static = (mixture.clone(), clean.clone())
        self.optimizer.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            enhanced = self.model(self._static[0])
            loss = self.loss_function(self._static[1], enhanced)
            loss.backward()
            self.optimizer.step()               # (host side of step(): state["step"] += 1 happened once, here)
            self._static_loss = loss.detach()
        self._graph = graph
        self._graph_sig = self.optimizer.hyper_signature()
        # capture does not execute: undo the host-side count of the captured call; every replay counts itself
        self.optimizer.advance_host_step(-1)

    def _step(self, mixture, clean):
        if not self.use_graph:
            return self._eager_step(mixture, clean)
        if self._graph is not None and self.optimizer.hyper_signature() != self._graph_sig:
            # lr / betas / eps / grad_scale are kernel arguments of the captured Adam step: a scheduler, a manual decay or a
            # load_state_dict with another lr would be ignored by the replay - capture again with the new values.  A schedule that
            # changes them every step (or every few steps) would spend its time in capture + instantiate (far slower than eager):
            # after MAX_RECAPTURES graphs in a row that did not earn their capture back (see the constants), eager launches.
            self._graph = None
            earned = self._replays >= MIN_REPLAYS and self._replays * REPLAY_CREDIT_S >= self._capture_s
            self._recaptures = 0 if earned else self._recaptures + 1
            self._replays = 0
            if self._recaptures >= MAX_RECAPTURES:
                import warnings
                warnings.warn("Trainer: the optimiser's hyper-parameters change every few steps (a per-step LR schedule?); the captured "
                              "step graph would be rebuilt each time - falling back to eager launches", RuntimeWarning)
                self.use_graph = False
                self._fall_back_to_host_step()
                return self._eager_step(mixture, clean)
        if self._graph is None:
            if self._eager_steps < GRAPH_WARMUP_STEPS:
                self._eager_steps += 1
                return self._eager_step(mixture, clean)
            import time
            t0 = time.perf_counter()
            self._capture(mixture, clean)
            self._capture_s = time.perf_counter() - t0
        if mixture.shape != self._static[0].shape:          # a ragged last batch: eager (same kernels, same numbers)
            loss = self._eager_step(mixture, clean)
            return loss
        self._static[0].copy_(mixture, non_blocking=True)
        self._static[1].copy_(clean, non_blocking=True)
        self._graph.replay()
        self._replays += 1
        self.optimizer.advance_host_step(1)
        return self._static_loss

    def _fall_back_to_host_step(self):
        """Eager from here on: the optimiser counts its steps on the host again (its state["step"] has been kept in step with the
        device counter by advance_host_step; the device counter is dropped and would be re-seeded if the graph ever came back)."""
        self.optimizer.device_step = False
        self.optimizer._dev = {}

    def _train_epoch(self, epoch):
        sampler = getattr(self.train_data_loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        loss_total = torch.zeros((), device=self.device)
        n = 0
        for mixture, clean, _name in self.train_data_loader:
            mixture = mixture.to(self.device, non_blocking=True)
            clean = clean.to(self.device, non_blocking=True)
            loss_total += self._step(mixture, clean)
            n += 1
        mean = (loss_total / max(n, 1)).item()          # the only device->host sync of the epoch
        self.epoch_losses.append(mean)
        return mean

    def train(self):
        for epoch in range(self.start_epoch, self.epochs + 1):
            self.model.train()
            self._train_epoch(epoch)
            if self.save_checkpoint_interval != 0 and epoch % self.save_checkpoint_interval == 0:
                self._save_checkpoint(epoch)
        if self._own_pg and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
