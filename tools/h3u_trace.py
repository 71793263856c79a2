"""Stage timeline of conv_h3u_kernel (measurement build: tools/_lib_trace.so, -DWUNET_H3U_TRACE): per block, the MFMA waves' and the loader
waves' arrival at / release from every stage barrier (shader clock, 100 MHz steps of s_memtime).  Prints who waits for whom."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WUNET_LIB_PATH"] = os.path.join(ROOT, "tools", "_lib_trace.so")
os.environ.setdefault("WUNET_H3U", "8192,0")
PKG = "wave-u-net-for-speech-enhancement_amd"
pkg = importlib.import_module(PKG)
eng = importlib.import_module(PKG + ".engine").default_engine()
dev = torch.device("cuda:0")
m = pkg.Model(n_layers=12, channels_interval=24).to(dev).eval()
x = torch.randn(64, 1, 16384, device=dev)
buf = torch.zeros((1 << 20) + 256 * 2 * 128 * 2 + 256 * 128 * 4 + 1024, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    eng.lib.wunet_debug_set_conv_trace(ctypes.c_void_p(buf.data_ptr()))
    m(x)
    torch.cuda.synchronize()
    eng.lib.wunet_debug_set_conv_trace(None)
t = buf[(1 << 20):(1 << 20) + 256 * 2 * 128 * 2].cpu().numpy().reshape(256, 2, 128, 2).astype(np.float64)
ns = 48                                    # stages per block of the last launch (decoder.11: 16 items x 3 chunks)
ok = t[:, 0, :ns, 1] > 0
print("blocks with stamps", int(ok.all(axis=1).sum()))
cyc = 1.0                                  # (s_memtime ticks: core clock cycles here, ~2.1 per cyc)
arr_c, rel_c, arr_l, rel_l = t[:, 0, :ns, 0], t[:, 0, :ns, 1], t[:, 1, :ns, 0], t[:, 1, :ns, 1]
stage = np.diff(rel_c, axis=1) * cyc
print("stage time cyc: mean %.0f median %.0f p10 %.0f p90 %.0f" % (stage.mean(), np.median(stage), np.percentile(stage, 10), np.percentile(stage, 90)))
wc = (rel_c - arr_c)[:, 1:] * cyc
wl = (rel_l - arr_l)[:, 1:ns - 1] * cyc
print("MFMA waves wait at the barrier cyc: mean %.0f median %.0f" % (wc.mean(), np.median(wc)))
print("loader waves wait at the barrier cyc: mean %.0f median %.0f" % (wl.mean(), np.median(wl)))
busy_c = (arr_c[:, 1:] - rel_c[:, :-1]) * cyc
busy_l = (arr_l[:, 1:ns - 1] - rel_l[:, :ns - 2]) * cyc
print("MFMA waves busy per stage cyc: mean %.0f median %.0f" % (busy_c.mean(), np.median(busy_c)))
print("loader waves busy per stage cyc: mean %.0f median %.0f" % (busy_l.mean(), np.median(busy_l)))
by_ch = [busy_l[:, (np.arange(busy_l.shape[1]) % 3) == k].mean() for k in range(3)]
print("loader busy by chunk of the tile being prepared (stage %% 3):", ["%.0f" % v for v in by_ch])
by_chc = [busy_c[:, (np.arange(busy_c.shape[1]) % 3) == k].mean() for k in range(3)]
print("MFMA busy by stage %% 3:", ["%.0f" % v for v in by_chc])
sub = buf[(1 << 20) + 256 * 2 * 128 * 2:(1 << 20) + 256 * 2 * 128 * 2 + 256 * 128 * 4].cpu().numpy().reshape(256, 128, 4).astype(np.float64)[:, 1:ns - 2]
rl = rel_l[:, 1:ns - 2]
print("loader stage, cycles: barrier release -> conversion entry (W DMA issue + prefetch issue) %.0f | coefficients %.0f | arithmetic %.0f | split + LDS writes %.0f | last columns + end %.0f" % (
    (sub[:, :, 0] - rl).mean(), (sub[:, :, 1] - sub[:, :, 0]).mean(), (sub[:, :, 2] - sub[:, :, 1]).mean(), (sub[:, :, 3] - sub[:, :, 2]).mean(),
    (arr_l[:, 2:ns - 1] - sub[:, :, 3]).mean()))
print("kernel span us (first release -> last arrival): %.1f" % ((arr_c.max() - rel_c[:, 0].min()) * cyc / 1e3))
