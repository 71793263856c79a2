"""the data input path either side of the hot loop.

Two pieces, both float32 mono like the reference's items:

* `Dataset` - drop-in for the reference's dataset/waveform_dataset.py:10-67 (same constructor arguments, same list
  file format "<noisy path><space><clean path>", same item contract `(mixture[1,T], clean[1,T], filename)`, same
  aligned random crop as util/utils.py:101-113 including its use of `np.random.randint`, so a seeded run crops the
  same windows).  The reference decodes with librosa.load(sr=None); this image has no audio library, so RIFF/WAVE
  files are parsed here (PCM 8/16/24/32 bit and IEEE float 32/64, any channel count -> mono mean, no resampling -
  what librosa.load(sr=None, mono=True) returns for such files).

      "train_dataset": {"module": "wave-u-net-for-speech-enhancement_amd.waveform_dataset", "main": "Dataset",
                        "args": {"dataset": "~/train.txt", "limit": null, "offset": 0, "sample_length": 16384, "mode": "train"}}

* `pack_shard` + `ShardLoader` - the MI355X-first form of the same path ("pre-decoded memory-mapped
  float32 shards + on-GPU crop").  Per-item librosa decoding in 40 worker processes cannot feed >50 k frames/s; a shard
  is the whole list decoded ONCE into two flat float32 files (noisy, clean) plus an index, memory-mapped, uploaded to HBM
  once (a 100-hour 16 kHz corpus is 46 GB of the 288 GB) and every batch is one gather on the GPU: random item, random
  aligned start, `[B,1,sample_length]` mixture and clean, no host work and no H2D copy per step.  Iterating it yields
  the DataLoader batch contract `(mixture, clean, names)` the reference trainer loop consumes (trainer/trainer.py:30).
"""
import json
import os
import struct

import numpy as np
import torch
from torch.utils import data


# ---------------------------------------------------------------------------------------------- WAV decoding
def read_wav(path):
    """float32 mono samples and the sample rate of a RIFF/WAVE file (what librosa.load(path, sr=None) returns)."""
    with open(os.path.abspath(os.path.expanduser(path)), "rb") as f:
        blob = f.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:           # WAVE_FORMAT_EXTENSIBLE: the real tag leads the GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif tag == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, channels, rate, _, _, bits = fmt
    if tag == 1:                                               # integer PCM
        if bits == 8:
            x = (np.frombuffer(payload, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(payload[:len(payload) // 2 * 2], "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(payload[:len(payload) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (np.frombuffer(payload[:len(payload) // 4 * 4], "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:                                             # IEEE float
        if bits not in (32, 64):
            raise ValueError(f"{path}: unsupported float width {bits}")
        x = np.frombuffer(payload[:len(payload) // (bits // 8) * (bits // 8)], "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    if channels > 1:
        x = x[:len(x) // channels * channels].reshape(-1, channels).mean(axis=1).astype(np.float32)
    return np.ascontiguousarray(x), rate


# ---------------------------------------------------------------------------------------------- the list file and the item contract
class PairList:
    """The reference's list file - one "<noisy path><space><clean path>" per line (dataset/waveform_dataset.py:15-29), `offset` lines
    skipped, then at most `limit` kept - parsed once into records.  Both readers of the format use it: `Dataset` (decode per item,
    the reference's contract) and `pack_shard` (decode once into a shard)."""

    def __init__(self, list_file, limit=None, offset=0):
        with open(os.path.abspath(os.path.expanduser(list_file)), "r") as f:
            rows = [ln.rstrip("\n") for ln in f][offset:]
        self.rows = rows[:limit] if limit else rows

    def __len__(self):
        return len(self.rows)

    def pair(self, k):
        """(noisy path, clean path, item name = the noisy file's stem) of record k."""
        noisy, clean = self.rows[k].split(" ")
        return noisy, clean, os.path.splitext(os.path.basename(noisy))[0]

    def decode(self, k):
        """Both signals of record k as float32 mono + their sample rates + the item name."""
        noisy, clean, name = self.pair(k)
        (x, rx), (y, ry) = read_wav(noisy), read_wav(clean)
        return x, y, rx, ry, name


def aligned_window_start(n_a, n_b, sample_length):
    """First sample of the ONE random window both signals are cut at (util/utils.py:101-113).  The draw is `np.random.randint` over
    the n - L + 1 valid starts, as in the reference: a run seeded with np.random.seed crops the windows the reference would (the
    device-side form of the same draw is ShardLoader.draw).  The two assertion texts are the reference's."""
    assert n_a == n_b, "Inconsistent dataset length, unable to sampling"
    assert n_a >= sample_length, f"len(data_a) is {n_a}, sample_length is {sample_length}."
    return int(np.random.randint(n_a - sample_length + 1))


class Dataset(data.Dataset):
    """Plugin-compatible with the reference's dataset/waveform_dataset.py:10-67: constructor arguments, `mode` check, and the item
    `(mixture [1, T], clean [1, T], filename)` - a random aligned window of `sample_length` in "train" mode, the whole file in
    "validation" mode."""
    MODES = ("train", "validation")

    def __init__(self, dataset, limit=None, offset=0, sample_length=16384, mode="train"):
        super().__init__()
        assert mode in self.MODES, "Mode must be one of 'train' or 'validation'."
        self.pairs = PairList(dataset, limit, offset)
        self.sample_length, self.mode = sample_length, mode

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, item):
        mixture, clean, _, _, name = self.pairs.decode(item)
        if self.mode == "train":
            s = aligned_window_start(len(mixture), len(clean), self.sample_length)
            mixture, clean = mixture[s:s + self.sample_length], clean[s:s + self.sample_length]
        return mixture[None, :], clean[None, :], name


# ---------------------------------------------------------------------------------------------- shards
def pack_shard(dataset, prefix, limit=None, offset=0):
    """Decode every pair of the list file once into `<prefix>.noisy.f32`, `<prefix>.clean.f32` (flat float32) and
    `<prefix>.index.json` ({"names", "starts", "lengths", "sample_rate"}).  Returns the number of items."""
    pairs = PairList(dataset, limit, offset)
    names, starts, lengths, rate, pos = [], [], [], None, 0
    with open(prefix + ".noisy.f32", "wb") as fn, open(prefix + ".clean.f32", "wb") as fc:
        for k in range(len(pairs)):
            mixture, clean, r0, r1, name = pairs.decode(k)
            assert len(mixture) == len(clean), f"Inconsistent dataset length: {pairs.pair(k)[0]}"
            assert r0 == r1 and (rate is None or rate == r0), f"mixed sample rates: {pairs.pair(k)[0]}"
            rate = r0
            fn.write(mixture.tobytes())
            fc.write(clean.tobytes())
            names.append(name)
            starts.append(pos)
            lengths.append(len(mixture))
            pos += len(mixture)
    with open(prefix + ".index.json", "w") as f:
        json.dump({"names": names, "starts": starts, "lengths": lengths, "sample_rate": rate}, f)
    return len(names)


class ShardLoader:
    """Batches of aligned random crops cut on the device from a shard that lives in device memory.

    for mixture, clean, names in ShardLoader(prefix, batch_size=64, device="cuda:0"): ...   # [B,1,L] float32 each
    Items shorter than `sample_length` are never drawn (the reference asserts on them).  Every rank of a data-parallel
    job passes its own `seed` (e.g. seed + rank) and draws independently, like shuffled per-rank loaders."""

    def __init__(self, prefix, batch_size, sample_length=16384, device="cpu", seed=0, steps_per_epoch=None, engine=None):
        idx = json.load(open(prefix + ".index.json"))
        self.names = idx["names"]
        starts = np.asarray(idx["starts"], np.int64)
        lengths = np.asarray(idx["lengths"], np.int64)
        usable = np.nonzero(lengths >= sample_length)[0]
        if len(usable) == 0:
            raise ValueError("no item is at least sample_length long")
        self.device = torch.device(device)
        total = int(starts[-1] + lengths[-1]) if len(starts) else 0
        self.noisy = self._resident(prefix + ".noisy.f32", total)                       # one upload; stays resident
        self.clean = self._resident(prefix + ".clean.f32", total)
        self.usable = torch.from_numpy(usable)
        self.starts = torch.from_numpy(starts)
        self.spare = torch.from_numpy(lengths - sample_length + 1)                      # number of valid window starts per item
        self.batch_size, self.sample_length = batch_size, sample_length
        self.steps = steps_per_epoch if steps_per_epoch is not None else max(1, len(usable) // batch_size)
        self.gen = torch.Generator().manual_seed(seed)
        self.engine = engine                     # tests inject the emulator engine; a GPU loader takes the default (HIP) engine

    UPLOAD_CHUNK = 1 << 26          # floats per staged piece (256 MB): the corpus is never copied into host RAM as a whole

    def _resident(self, path, total):
        """The flat float32 file as a tensor on self.device.  The file is memory-mapped (copy-on-write, so torch gets a writable
        view without touching the pages); a CPU loader uses the mapping itself, a GPU loader uploads it piece by piece."""
        mm = np.memmap(path, np.float32, "c", shape=(total,))
        if self.device.type == "cpu":
            return torch.from_numpy(mm)
        out = torch.empty(total, dtype=torch.float32, device=self.device)
        for a in range(0, total, self.UPLOAD_CHUNK):
            b = min(total, a + self.UPLOAD_CHUNK)
            out[a:b].copy_(torch.from_numpy(mm[a:b]))
        return out

    def __len__(self):
        return self.steps

    def draw(self):
        """(item ids, window starts inside the flat arrays) of one batch - host-side integers only."""
        pick = self.usable[torch.randint(len(self.usable), (self.batch_size,), generator=self.gen)]
        inner = (torch.rand(self.batch_size, generator=self.gen, dtype=torch.float64) * self.spare[pick].double()).long()
        inner = torch.minimum(inner, self.spare[pick] - 1)
        return pick, self.starts[pick] + inner

    def crop(self, first):
        """The batch of aligned windows that start at `first` (host int64 [B], offsets into the flat arrays): mixture and clean
        [B, 1, L].  On a GPU this is ONE launch of the HIP crop kernel behind the C ABI (wunet_crop_windows, csrc/wunet_ops.cpp);
        a host loader (device="cpu": the reference's DataLoader worker side) slices the memory map."""
        B, L = first.numel(), self.sample_length
        if self.device.type != "cuda" and self.engine is None:
            index = first[:, None] + torch.arange(L)[None, :]
            return self.noisy[index].unsqueeze(1), self.clean[index].unsqueeze(1)
        if self.engine is None:
            from .engine import default_engine
            self.engine = default_engine()
        mixture = torch.empty(B, 1, L, dtype=torch.float32, device=self.device)
        clean = torch.empty(B, 1, L, dtype=torch.float32, device=self.device)
        starts = first.to(self.device, non_blocking=True)
        self.engine.crop_windows(self.noisy, self.clean, starts, mixture, clean)
        return mixture, clean

    def __iter__(self):
        for _ in range(self.steps):
            pick, first = self.draw()
            mixture, clean = self.crop(first)
            yield mixture, clean, [self.names[i] for i in pick.tolist()]
