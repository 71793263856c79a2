"""Synthetic dataset plugin: the same item contract as the reference's
dataset/waveform_dataset.py:56-67 - `(mixture[1,T], clean[1,T], name)` float32 - without audio files:
clean ~ U(-1,1) (the range of librosa.load floats), mixture = clean + 0.1 N(0,1).

    "train_dataset": {"module": "wave-u-net-for-speech-enhancement_amd.dataset", "main": "Dataset",
                      "args": {"n_items": 4096, "sample_length": 16384, "seed": 0}}
"""
import torch
from torch.utils import data


class Dataset(data.Dataset):
    def __init__(self, n_items=1024, sample_length=16384, seed=0, mode="train"):
        assert mode in ("train", "validation"), "Mode must be one of 'train' or 'validation'."
        self.n_items, self.sample_length, self.seed = n_items, sample_length, seed

    def __len__(self):
        return self.n_items

    def __getitem__(self, item):
        g = torch.Generator().manual_seed(self.seed * 1000003 + item)
        clean = torch.rand(1, self.sample_length, generator=g) * 2 - 1
        mixture = clean + 0.1 * torch.randn(1, self.sample_length, generator=g)
        return mixture, clean, f"synthetic_{item:06d}"
