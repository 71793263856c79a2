"""Measurements of the "next" rows of SURVEY.md section 8(f) on one MI355X (GPU box): f1 the fused Adam step against torch.optim.Adam,
f3 chunked inference (one batched forward) against the reference's loop of batch-1 forwards with a copy per chunk, f4 the shard
loader with the on-device crop against the reference-contract Dataset behind a DataLoader.  (f2, the step driver: tools/host_phases.py
and bench.py --graph.)  Synthetic data; prints a table."""
import importlib, os, sys, tempfile, time, wave
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

PKG = "wave-u-net-for-speech-enhancement_amd"
pkg = importlib.import_module(PKG)
optim = importlib.import_module(PKG + ".optim")
inference = importlib.import_module(PKG + ".inference")
wd = importlib.import_module(PKG + ".waveform_dataset")
dev = torch.device("cuda:0")


_ballast = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def gpu_ms(fn, reps):
    """(GPU ms, host ms) per call.  The GPU figure is taken behind ~10 ms of queued fills, so that the host is ahead of the GPU for
    the whole timed region (a loop of calls that the host issues slower than the GPU runs them would measure the host)."""
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(40):
        _ballast.fill_(1.0)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps, host


# ---------------------------------------------------------------- f1
torch.manual_seed(0)
m = pkg.Model().to(dev).train()
n_par = sum(p.numel() for p in m.parameters())
for p in m.parameters():
    p.grad = torch.randn_like(p) * 1e-3
fa = optim.FusedAdam(m.parameters(), lr=1e-4)
g_f, h_f = gpu_ms(fa.step, 20)
m2 = pkg.Model().to(dev).train()
for p in m2.parameters():
    p.grad = torch.randn_like(p) * 1e-3
ta = torch.optim.Adam(m2.parameters(), lr=1e-4)
g_t, h_t = gpu_ms(ta.step, 20)
byt = 7 * 4 * n_par
print("f1  Adam step over the %d tensors / %.1f M parameters of the 12-level net (reads p, g, m, v; writes p, m, v = %.0f MB)" % (len(list(m.parameters())), n_par / 1e6, byt / 1e6))
print("    FusedAdam (optim.py, one launch per 64 tensors)   GPU %.3f ms = %.2f TB/s   host %.3f ms" % (g_f, byt / g_f / 1e9, h_f))
print("    torch.optim.Adam (foreach)                         GPU %.3f ms = %.2f TB/s   host %.3f ms" % (g_t, byt / g_t / 1e9, h_t))

# ---------------------------------------------------------------- f3
m.eval()
T = 600 * 16000                                   # a ten-minute recording at 16 kHz
mix = torch.randn(1, 1, T, device=dev) * 0.1
nch = (T + 16383) // 16384


def batched():
    return inference.enhance(m, mix, to_host=True)


def reference_loop():                              # enhancement.py:57-69 on the same model: batch 1, a copy per chunk
    pad = (-T) % 16384
    x = torch.cat([mix, torch.zeros(1, 1, pad, device=dev)], dim=-1)
    outs = []
    with torch.no_grad():
        for c in torch.split(x, 16384, dim=-1):
            outs.append(m(c.contiguous()).detach().cpu())
    return torch.cat(outs, dim=-1)[:, :, :T]


a = batched(); b = reference_loop()
same = float((a - b).abs().max())
t0 = time.perf_counter(); batched(); torch.cuda.synchronize(); tb = time.perf_counter() - t0
t0 = time.perf_counter(); reference_loop(); torch.cuda.synchronize(); tl = time.perf_counter() - t0
print("f3  enhancing a %d-sample recording (%d chunks of 16384), result on the host; max |batched - loop| = %.1e" % (T, nch, same))
print("    inference.enhance (equal slabs of <= 256 chunks, async copies into one pinned buffer) %.1f ms = %.0f chunks/s" % (tb * 1e3, nch / tb))
print("    the reference's loop (batch 1, a copy per chunk)   %.1f ms = %.0f chunks/s   -> %.1fx" % (tl * 1e3, nch / tl, tl / tb))

# ---------------------------------------------------------------- f4
tmp = tempfile.mkdtemp(prefix="wunet_f4_")
rng = np.random.default_rng(0)
n_items, n_samp = 256, 64000
lines = []
for i in range(n_items):
    pair = []
    for kind in ("noisy", "clean"):
        path = os.path.join(tmp, "%s_%04d.wav" % (kind, i))
        with wave.open(path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((rng.standard_normal(n_samp) * 3000).astype("<i2").tobytes())
        pair.append(path)
    lines.append(" ".join(pair))
lst = os.path.join(tmp, "list.txt")
open(lst, "w").write("\n".join(lines) + "\n")
print("f4  %d pairs of %d-sample 16-bit WAV files; batches of 64 aligned random 16384-sample crops" % (n_items, n_samp))
for workers in (0, 8):
    ds = wd.Dataset(lst, sample_length=16384, mode="train")
    dl = torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True, num_workers=workers, pin_memory=True, persistent_workers=workers > 0)
    for _ in dl:                                   # (workers started, files in the page cache)
        pass
    n = 0
    t0 = time.perf_counter()
    for _ in range(10):
        for mixture, clean, names in dl:
            mixture = mixture.to(dev, non_blocking=True); clean = clean.to(dev, non_blocking=True)
            n += mixture.shape[0]
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("    reference-contract Dataset + DataLoader(num_workers=%d), decode + crop per item + upload   %.0f frames/s" % (workers, n / t))
t0 = time.perf_counter()
prefix = os.path.join(tmp, "shard")
wd.pack_shard(lst, prefix)
tp = time.perf_counter() - t0
ld = wd.ShardLoader(prefix, batch_size=64, sample_length=16384, device="cuda:0", steps_per_epoch=500)
torch.cuda.synchronize()
n = 0
t0 = time.perf_counter()
for mixture, clean, names in ld:
    n += mixture.shape[0]
torch.cuda.synchronize()
t = time.perf_counter() - t0
first = ld.draw()[1]
gk, hk = gpu_ms(lambda: ld.crop(first), 100)
print("    pack_shard once: %.2f s; ShardLoader (shard resident in HBM, one crop launch per batch)           %.0f frames/s" % (tp, n / t))
print("    the crop launch alone: GPU %.4f ms per batch of 64 (%.2f TB/s of 2 x 2 x 4 MB), host %.3f ms" % (gk, 4 * 64 * 16384 * 4 / gk / 1e9, hk))
