"""train-step driver, selectable as the reference's trainer plugin

    "trainer": {"module": "wave-u-net-for-speech-enhancement_amd.trainer", "main": "Trainer", "epochs": ..., ...}

Same constructor as the reference's Trainer (/root/reference/trainer/trainer.py:13-25, called from train.py:41-49)
and the same `.train()` entry (train.py:51), covering the hot loop only (trainer/trainer.py:27-43,
base_trainer.py:187-197): validation, TensorBoard and metrics are out of scope.
Differences from the reference loop, all on the f2 list:
  * one process per GPU (torchrun); the RCCL gradient all-reduce is attached to the model instead of
    nn.DataParallel (base_trainer.py:26-27);
  * host->device copies are non_blocking from pinned memory (train.py:20 sets pin_memory but copies synchronously);
  * the loss is accumulated on the device and read back once per epoch (trainer.py:40 syncs every step);
  * checkpoints keep the reference's names and keys (base_trainer.py:83-124) so either side can resume.
"""
import os
from pathlib import Path

import torch
import torch.distributed as dist

from .parallel import GradSync


class Trainer:
    def __init__(self, config, resume, model, loss_function, optimizer, train_dataloader, validation_dataloader=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
        self.model = model.to(self.device)
        if self.world > 1:
            self.model.grad_sync = GradSync()
        self.loss_function = loss_function
        self.optimizer = optimizer
        self.train_data_loader = train_dataloader
        self.validation_data_loader = validation_dataloader      # accepted for signature compatibility, unused
        tcfg = config["trainer"]
        self.epochs = tcfg["epochs"]
        self.save_checkpoint_interval = tcfg.get("save_checkpoint_interval", 0)
        self.start_epoch = 1
        self.best_score = float("-inf")
        root = Path(os.path.expanduser(config.get("root_dir", "."))).absolute() / config.get("experiment_name", "exp")
        self.checkpoints_dir = root / "checkpoints"
        self.epoch_losses = []
        if resume:
            self._resume_checkpoint()

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

    # ---- the hot loop (trainer/trainer.py:27-43)
    def _train_epoch(self, epoch):
        loss_total = torch.zeros((), device=self.device)
        n = 0
        for mixture, clean, _name in self.train_data_loader:
            mixture = mixture.to(self.device, non_blocking=True)
            clean = clean.to(self.device, non_blocking=True)
            self.optimizer.zero_grad(set_to_none=True)
            enhanced = self.model(mixture)
            loss = self.loss_function(clean, enhanced)
            loss.backward()
            self.optimizer.step()
            loss_total += loss.detach()
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
