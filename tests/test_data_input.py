"""CPU tests of SURVEY.md §8(f4): the waveform dataset plugin (reference dataset/waveform_dataset.py semantics without
librosa) and the pre-decoded shard + on-device crop loader."""
import importlib
import struct
import wave

import numpy as np
import pytest
import torch

from conftest import PKG_NAME


@pytest.fixture(scope="module")
def wd():
    return importlib.import_module(PKG_NAME + ".waveform_dataset")


def _write_pcm16(path, x, rate=16000, channels=1):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.clip(x, -1, 1 - 1 / 32768) * 32768).astype("<i2").tobytes())


def _write_float32(path, x, rate=16000):
    body = x.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, 1, rate, rate * 4, 4, 32)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"LIST" + struct.pack("<I", 3) + b"abc" + b"\0")          # an odd-sized chunk the parser has to skip
        f.write(b"data" + struct.pack("<I", len(body)) + body)


def _corpus(tmp_path, n=5, seed=0):
    rng = np.random.default_rng(seed)
    lines, items = [], []
    for i in range(n):
        T = int(rng.integers(300, 700))
        clean = (rng.random(T) * 2 - 1).astype(np.float32) * 0.9
        noisy = np.clip(clean + 0.05 * rng.standard_normal(T).astype(np.float32), -0.99, 0.99)
        pn, pc = tmp_path / f"n{i}.wav", tmp_path / f"c{i}.wav"
        _write_pcm16(pn, noisy)
        _write_pcm16(pc, clean)
        q = lambda v: (np.clip(v, -1, 1 - 1 / 32768) * 32768).astype("<i2").astype(np.float32) / 32768.0
        items.append((q(noisy), q(clean), f"n{i}"))
        lines.append(f"{pn} {pc}")
    lst = tmp_path / "train.txt"
    lst.write_text("\n".join(lines) + "\n")
    return lst, items


def test_read_wav_formats(wd, tmp_path):
    x = np.linspace(-0.9, 0.9, 101).astype(np.float32)
    _write_pcm16(tmp_path / "a.wav", x)
    y, rate = wd.read_wav(tmp_path / "a.wav")
    assert rate == 16000 and y.dtype == np.float32 and np.abs(y - x).max() <= 1 / 32768
    _write_float32(tmp_path / "b.wav", x, rate=8000)
    y, rate = wd.read_wav(tmp_path / "b.wav")
    assert rate == 8000 and np.array_equal(y, x)
    st = np.stack([x, -x], axis=1).reshape(-1)                             # stereo -> mono mean (librosa mono=True)
    _write_pcm16(tmp_path / "c.wav", st, channels=2)
    y, _ = wd.read_wav(tmp_path / "c.wav")
    assert len(y) == len(x) and np.abs(y).max() <= 1 / 32768
    (tmp_path / "d.wav").write_bytes(b"not a wav file at all")
    with pytest.raises(ValueError):
        wd.read_wav(tmp_path / "d.wav")


def test_dataset_matches_reference_semantics(wd, tmp_path):
    lst, items = _corpus(tmp_path)
    ds = wd.Dataset(str(lst), limit=4, offset=1, sample_length=256, mode="train")
    assert len(ds) == 4
    np.random.seed(7)
    got = [ds[i] for i in range(len(ds))]
    np.random.seed(7)                                                      # the reference's crop draws np.random.randint the same way
    for (mix, cl, name), (noisy, clean, nm) in zip(got, items[1:5]):
        start = np.random.randint(len(noisy) - 256 + 1)
        assert mix.shape == (1, 256) and cl.shape == (1, 256) and mix.dtype == np.float32 and name == nm
        assert np.array_equal(mix[0], noisy[start:start + 256]) and np.array_equal(cl[0], clean[start:start + 256])
    val = wd.Dataset(str(lst), mode="validation")
    mix, cl, name = val[0]
    assert mix.shape == (1, len(items[0][0])) and np.array_equal(cl[0], items[0][1])
    with pytest.raises(AssertionError):
        wd.Dataset(str(lst), mode="test")
    with pytest.raises(AssertionError):
        wd.Dataset(str(lst), sample_length=100000)[0]                      # shorter than sample_length: the reference asserts too


def _emu_engine():
    import importlib
    import emu_lib
    from conftest import PKG_NAME
    eng_mod = importlib.import_module(PKG_NAME + ".engine")
    lib_mod = importlib.import_module(PKG_NAME + "._lib")
    return eng_mod.Engine(lib=lib_mod.declare(emu_lib.lib()), host_memory=True)


@pytest.mark.parametrize("through_kernel", [False, True], ids=["host-slices", "crop-kernel"])
def test_shard_loader_crops_on_device(wd, tmp_path, through_kernel):
    """Every window of a batch is a window of the item it names, mixture and clean cut at the SAME start - the reference's
    sample_fixed_length_data_aligned (util/utils.py:101-113) - checked against the corpus as written, not against the loader.
    through_kernel: the batch is cut by crop_windows_kernel behind wunet_crop_windows (the GPU loader's path), run here on the
    emulator build of the same kernel source."""
    lst, items = _corpus(tmp_path, n=6, seed=3)
    prefix = str(tmp_path / "shard")
    assert wd.pack_shard(str(lst), prefix) == 6
    L = 320
    eng = _emu_engine() if through_kernel else None
    loader = wd.ShardLoader(prefix, batch_size=4, sample_length=L, device="cpu", seed=11, steps_per_epoch=3, engine=eng)
    by_name = {nm: (noisy, clean) for noisy, clean, nm in items}
    seen = 0
    for mixture, clean, names in loader:
        assert mixture.shape == (4, 1, L) and clean.shape == (4, 1, L) and mixture.dtype == torch.float32
        for b, nm in enumerate(names):
            noisy_src, clean_src = by_name[nm]
            assert len(noisy_src) >= L                                     # short items are never drawn
            m = mixture[b, 0].numpy()
            hits = [s for s in range(len(noisy_src) - L + 1) if np.array_equal(noisy_src[s:s + L], m)]
            assert len(hits) == 1
            assert np.array_equal(clean[b, 0].numpy(), clean_src[hits[0]:hits[0] + L])       # aligned with the mixture
        seen += 1
    assert seen == len(loader) == 3
    again = wd.ShardLoader(prefix, batch_size=4, sample_length=L, device="cpu", seed=11, steps_per_epoch=1, engine=eng)
    first = next(iter(wd.ShardLoader(prefix, batch_size=4, sample_length=L, device="cpu", seed=11, steps_per_epoch=1)))
    assert torch.equal(next(iter(again))[0], first[0])                     # seeded: reproducible, and both paths cut the same windows
    with pytest.raises(ValueError):
        wd.ShardLoader(prefix, batch_size=2, sample_length=100000)


def test_crop_kernel_edges(wd):
    """wunet_crop_windows on the emulator: ragged row length (not a multiple of 4: the scalar tail), a window at the very end of the
    shard, starts outside the shard clamped instead of read."""
    eng = _emu_engine()
    total, L, B = 1000, 37, 5
    flat_m = torch.arange(total, dtype=torch.float32)
    flat_c = -torch.arange(total, dtype=torch.float32)
    starts = torch.tensor([0, 1, total - L, total + 50, -7], dtype=torch.int64)
    mix, cl = torch.empty(B, 1, L), torch.empty(B, 1, L)
    eng.crop_windows(flat_m, flat_c, starts, mix, cl)
    for b, s0 in enumerate([0, 1, total - L, total - L, 0]):
        assert torch.equal(mix[b, 0], flat_m[s0:s0 + L]) and torch.equal(cl[b, 0], flat_c[s0:s0 + L]), b
