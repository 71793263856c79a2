#!/usr/bin/env python
"""Benchmark of the Wave-U-Net hot path on MI355X (contract: see the round brief).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one training step of the reference's hot loop (trainer/trainer.py:34-38) on one batch of
synthetic frames already resident in HBM: zero_grad, forward, smooth-L1 loss, backward (with the
bucketed RCCL gradient all-reduce when N>1), Adam step.  Metric: 16384-sample frames per second,
whole job.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch
import torch.distributed as dist

PKG = "wave-u-net-for-speech-enhancement_amd"
N_LAYERS, CI, FRAME = 12, 24, 16384
FWD_FLOP_PER_FRAME = 4.885e9
FWDBWD_FLOP_PER_FRAME = 14.643e9
FWDBWD_BYTES_PER_FRAME = 99.54e6
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak == fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (v_mfma_f32_16x16x32_f16)
# the fp16-split kernels spend THREE f16 MFMA passes per fp32-equivalent product (hi*hi + hi*lo + lo*hi), so their
# ceiling in algorithmic (fp32) flops is a third of the f16 peak
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0
SUSTAINED_F16_MFMA_TFLOPS = 2070.0     # measured here: a register-only MFMA loop, clock settled at the board's power limit (profiles/r3_power_probe.txt)


def net_flops_bytes(n, ci, frame):
    """Algorithmic forward flops and bytes per frame: sum over the 2n+2 convs of 2*Cin*Cout*k*L and
    4*(Cin+Cout)*L."""
    plan = importlib.import_module(PKG + ".plan")
    fl = by = 0.0
    shapes = plan.conv_layer_shapes(n, ci)
    for i, (c_in, c_out, k) in enumerate(shapes):
        L = frame >> (i if i <= n else 2 * n - i)
        fl += 2.0 * c_in * c_out * k * L
        by += 4.0 * (c_in + c_out) * L
    fl += 2.0 * (ci + 1) * frame
    by += 4.0 * (ci + 2) * frame
    return fl, by


def _code_only(text):
    """C / C++ source with comments removed and white space collapsed: what the compiler sees, up to spelling.  (String and character
    literals are walked so that a // or /* inside one is kept.)"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == "/" and i + 1 < n and text[i + 1] == "/":
            while i < n and text[i] != "\n":
                if text[i] == "\\" and i + 1 < n:          # a line comment continued by a backslash
                    i += 1
                i += 1
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def source_hash():
    """sha1 over the CODE of the kernel + host sources (comments and white space do not count: a comment edit after the PMC pass must
    not void the round's traffic figures - VERDICT r4 #10): measurements stored under profiles/ are stamped with it and refused when
    stale."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, PKG, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, PKG, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, "r", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def load_pmc_traffic(section=None):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside the bench): profiles/pmc_traffic.json,
    produced by tools/measure_round.sh + tools/collect_round.py, stamped with the source hash of the build it was measured on.
    section: None = the headline workload; "gemm_fp32" / "deep16_bf16" = the PMC passes of the extras (same file, own sub-object)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        pmc = json.load(open(path))
    except (OSError, ValueError):
        return {"kernels": None, "why": "profiles/pmc_traffic.json missing"}
    if pmc.get("source_hash") != source_hash():
        return {"kernels": None, "why": f"profiles/pmc_traffic.json is stale (measured on sources {pmc.get('source_hash')}, these are "
                                        f"{source_hash()}): refused"}
    if section is not None:
        sub = (pmc.get("sections") or {}).get(section)
        if sub is None:
            return {"kernels": None, "why": f"profiles/pmc_traffic.json holds no PMC pass of '{section}'"}
        pmc = dict(sub)
    pmc["why"] = ("profiles/pmc_traffic.json (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, FETCH "
                  "doubled per the gfx950 note of MI355X_MICROARCH.md; measured on these sources)")
    return pmc


def kernel_rows(lib, fn, nprof):
    """Per-kernel durations of `nprof` calls of fn: HIP events recorded by the library on the launch stream around every kernel it
    annotates (events perturb the launch stream slightly, so headline values are taken without them)."""
    lib.wunet_profile_enable(1)
    for _ in range(nprof):
        fn()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.wunet_profile_collect(buf, len(buf))
    lib.wunet_profile_enable(0)
    rows = []
    for line in buf.value.decode().strip().splitlines():
        name, n, ms, fl, by = line.split("\t")
        rows.append({"kernel": name, "launches": int(n), "ms": float(ms), "flops": float(fl), "bytes": float(by)})
    rows.sort(key=lambda r: -r["ms"])
    return rows


def gemm_peak(kernel):
    if "bf16" in kernel:
        return PEAK_F16_MFMA_TFLOPS, "2500 TFLOP/s dense bf16 MFMA"
    if "_h3" in kernel:
        return PEAK_SPLIT_TFLOPS, "2500 TFLOP/s dense f16 MFMA / 3 passes per fp32-equivalent product"
    return PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA (v_mfma_f32_16x16x4_f32)"


def roofline_of(rows, nprof, pmc, step_algorithmic_bytes):
    """The roofline object of the contract for the GEMM kernel that takes the most time in `rows` (+ the HBM-side view of the same
    kernel, the other GEMM kernels, the HBM-bound kernels and the whole-step traffic from the PMC file)."""
    gemm_rows = [r for r in rows if r["flops"] > 0 and ("mfma" in r["kernel"] or "_h3" in r["kernel"])]
    mem_rows = [r for r in rows if r not in gemm_rows]
    if not gemm_rows:
        return None
    top = gemm_rows[0]
    avg_ms = top["ms"] / top["launches"]
    achieved = top["flops"] / top["launches"] / (avg_ms * 1e-3) / 1e12
    mfma_ms = sum(r["ms"] for r in gemm_rows) / nprof
    traffic, traffic_src = None, pmc["why"]
    if pmc["kernels"] is not None and top["kernel"] in pmc["kernels"]:
        traffic = pmc["kernels"][top["kernel"]]["hbm_bytes_per_launch"]
    peak, peak_note = gemm_peak(top["kernel"])
    # the HBM-bound kernels of the shallow levels (north_star: "rocprof-reported HBM GB/s for the memory-bound shallow
    # levels"): algorithmic bytes / HIP-event time, same pass
    groups = {}
    for r in mem_rows:
        key = r["kernel"].split("<")[0]
        g = groups.setdefault(key, {"kernel": key, "ms": 0.0, "bytes": 0.0, "launches": 0})
        g["ms"] += r["ms"]; g["bytes"] += r["bytes"]; g["launches"] += r["launches"]
    memory_bound = [{"kernel": g["kernel"], "launches_per_step": g["launches"] / nprof, "ms_per_step": g["ms"] / nprof,
                     "algorithmic_GBps": g["bytes"] / (g["ms"] * 1e-3) / 1e9,
                     "frac_of_hbm_peak": g["bytes"] / (g["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS}
                    for g in sorted(groups.values(), key=lambda g: -g["ms"]) if g["ms"] > 0]
    whole = None
    if pmc["kernels"] is not None and pmc.get("whole_step_bytes"):
        whole = {"hbm_bytes_per_step": pmc["whole_step_bytes"], "algorithmic_bytes_per_step": step_algorithmic_bytes,
                 "ratio": pmc["whole_step_bytes"] / step_algorithmic_bytes}
    # the roof that binds THIS kernel: its algorithmic intensity against the ridge of the arithmetic it runs (a GEMM with <= 72 channels
    # on one side, or the bf16 levels of the deep variant, sits left of the ridge: the HBM roof is the one to read `frac` against)
    intensity = top["flops"] / max(top["bytes"], 1.0)
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    gbps = top["bytes"] / top["launches"] / (avg_ms * 1e-3) / 1e9
    hbm_bound = intensity < ridge
    return {"bound": "hbm" if hbm_bound else "mfma", "kernel": top["kernel"],
            "achieved": gbps if hbm_bound else achieved, "peak": PEAK_HBM_GBS if hbm_bound else peak,
            "peak_note": "8 TB/s HBM3E (MI355X_MICROARCH.md)" if hbm_bound else peak_note,
            "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": (gbps / PEAK_HBM_GBS) if hbm_bound else achieved / peak,
            # both views of the same kernel, whichever binds
            "mfma_view": {"achieved": achieved, "peak": peak, "peak_note": peak_note, "unit": "TFLOP/s", "frac": achieved / peak},
            "hbm_view": {"achieved": gbps, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBS},
            # beside the data-sheet roof: what a register-only loop of the same MFMA sustains on this part once the clock has settled at
            # the board's power limit (profiles/r3_power_probe.txt: 2.07 PFLOP/s f16 at 1316 W and 2.05 GHz) - f16 kernels only
            "power_limited_peak": (None if "_h3" not in top["kernel"] else
                                   {"value": SUSTAINED_F16_MFMA_TFLOPS / (1.0 if "bf16" in top["kernel"] else 3.0), "unit": "TFLOP/s",
                                    "frac": achieved / (SUSTAINED_F16_MFMA_TFLOPS / (1.0 if "bf16" in top["kernel"] else 3.0)),
                                    "source": "tools/power_probe.sh, profiles/r3_power_probe.txt"}),
            "traffic": traffic,
            "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
            "traffic_whole_step": whole,
            "algorithmic_bytes_per_launch": top["bytes"] / top["launches"],
            # the same kernel against the other roof: its algorithmic bytes / time, and where it sits relative to the ridge
            "algorithmic_GBps": top["bytes"] / top["launches"] / (avg_ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": top["bytes"] / top["launches"] / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "intensity_flop_per_byte": top["flops"] / max(top["bytes"], 1.0),
            "ridge_flop_per_byte": peak * 1e12 / (PEAK_HBM_GBS * 1e9),
            "avg_launch_ms": avg_ms, "launches_per_step": top["launches"] / nprof,
            "mfma_kernels_ms_per_step": mfma_ms,
            "all_mfma_kernels_achieved": sum(r["flops"] for r in gemm_rows) / nprof / (mfma_ms * 1e-3) / 1e12,
            "top5": [{"kernel": r["kernel"], "ms_per_step": r["ms"] / nprof, "launches_per_step": r["launches"] / nprof,
                      "tflops": r["flops"] / (r["ms"] * 1e-3) / 1e12, "algorithmic_bytes_per_step": r["bytes"] / nprof}
                     for r in gemm_rows[:(999 if os.environ.get("WUNET_BENCH_ALL") else 5)]],
            "memory_bound_kernels": memory_bound,
            "memory_bound_ms_per_step": sum(m["ms_per_step"] for m in memory_bound),
            "kernels_timed_per_step": sum(r["launches"] for r in rows) / nprof}


def timed(fn, warmup, steps, batch, what, median_steps=0):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r = {"what": what, "ms_per_step": dt * 1e3, "frames_per_s": batch / dt, "steps": steps, "warmup": warmup}
    if median_steps:
        # the same call timed step by step with device events: median and quartiles say whether the window above was disturbed
        per = []
        for _ in range(median_steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            per.append(e0.elapsed_time(e1))
        per.sort()
        r.update(ms_per_step_median=per[len(per) // 2], ms_per_step_p25=per[len(per) // 4], ms_per_step_p75=per[(3 * len(per)) // 4],
                 frames_per_s_at_median=batch / per[len(per) // 2] * 1e3)
    return r


def synthetic_batch(batch, device, seed, frame=FRAME):
    g = torch.Generator().manual_seed(seed)
    clean = torch.rand(batch, 1, frame, generator=g) * 2 - 1
    noisy = clean + 0.1 * torch.randn(batch, 1, frame, generator=g)
    return noisy.to(device), clean.to(device)


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(seconds_budget=25.0):
    """The reference's CPU path (the same ATen ops via oracle/torch_port.py, which is bit-identical to the
    imported reference - the reference itself cannot travel to the GPU box) on the host cores.  value = configs[0]: B=4 training steps
    with the `mse_loss` of config/train/train.json:27-31 + Adam(lr 1e-3, betas 0.9 / 0.999), 3 warm-up steps and the median of >= 10
    at the best thread count of a short sweep.  Beside it, at that thread count: the same step with smooth_l1 (configs[2]'s loss),
    eval forward at B=1 / B=4 (configs[1]) and the like-for-like B=64 forward and training step."""
    from oracle import plan, torch_port
    torch.manual_seed(0)
    sd = torch_port.state_to_torch(plan.golden_state(N_LAYERS, CI, 0), requires_grad=True)
    params = [v for k, v in sd.items() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999))
    noisy, clean = synthetic_batch(4, "cpu", 0)

    def one_step(loss_kind="mse", nz=None, cl=None):
        nz = noisy if nz is None else nz
        cl = clean if cl is None else cl
        t0 = time.perf_counter()
        opt.zero_grad()
        out = torch_port.forward(sd, nz, N_LAYERS, CI, True)
        loss = torch_port.loss_value(loss_kind, cl, out)
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    def median(ts):
        ts = sorted(ts)
        return ts[len(ts) // 2]

    # the host has far more cores than a batch-4 step can use: short sweep of the thread count (1 warm-up + 3 steps each), keep the best
    ncpu = os.cpu_count() or 1
    sweep = sorted({t for t in (8, 16, 32, 64, ncpu // 2) if 1 <= t <= ncpu})
    best = None
    t_start = time.perf_counter()
    tried = {}
    for nt in sweep:
        torch.set_num_threads(nt)
        one_step()
        t = median(one_step() for _ in range(3))
        tried[nt] = 4.0 / t
        if best is None or t < best[1]:
            best = (nt, t)
        if time.perf_counter() - t_start > seconds_budget * 0.4:
            break
    torch.set_num_threads(best[0])
    for _ in range(3):
        one_step()
    n_timed = 10
    t_mse = median(one_step("mse") for _ in range(n_timed))
    t_sl1 = median(one_step("smooth_l1") for _ in range(5))
    like = {"train_batch4_smooth_l1_frames_per_s": 4.0 / t_sl1}
    with torch.no_grad():
        sde = {k: v.detach() for k, v in sd.items()}
        for b in (1, 4):
            xb, _ = synthetic_batch(b, "cpu", 1)
            torch_port.forward(sde, xb, N_LAYERS, CI, False)
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                torch_port.forward(sde, xb, N_LAYERS, CI, False)
                ts.append(time.perf_counter() - t0)
            like[f"eval_forward_batch{b}_frames_per_s"] = b / median(ts)
    # batch 64 is 16 x the work per op: its own best of two thread counts (the B=4 optimum and 4 x that)
    n64, c64 = synthetic_batch(64, "cpu", 0)
    best64 = None
    for nt in sorted({best[0], min(ncpu, 4 * best[0])}):
        torch.set_num_threads(nt)
        one_step("smooth_l1", n64, c64)
        t = median(one_step("smooth_l1", n64, c64) for _ in range(3))
        if best64 is None or t < best64[1]:
            best64 = (nt, t)
    like["train_batch64_frames_per_s"] = 64.0 / best64[1]
    like["train_batch64_threads"] = best64[0]
    torch.set_num_threads(best64[0])
    with torch.no_grad():
        xb, _ = synthetic_batch(64, "cpu", 1)
        torch_port.forward(sde, xb, N_LAYERS, CI, False)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            torch_port.forward(sde, xb, N_LAYERS, CI, False)
            ts.append(time.perf_counter() - t0)
        like["eval_forward_batch64_frames_per_s"] = 64.0 / median(ts)
    return {"value": 4.0 / t_mse, "unit": "frames/s", "cores": best[0], "kind": "port", "cpu_model": cpu_model_string(), "cpu_count": ncpu,
            "sample": f"batch=4 x {FRAME}-sample frames, zero_grad + fwd + mse_loss + bwd + Adam (BASELINE.json configs[0], "
                      f"config/train/train.json), 3 warm-up steps, median of {n_timed} at the best of {list(tried)} threads "
                      f"(torch {torch.__version__} CPU ATen kernels = the reference's CUDA_VISIBLE_DEVICES=-1 path)",
            "frames_per_s_by_threads": tried,
            # like-for-like figures at the same thread count: configs[1] eval forward, configs[2] at B=64
            **like,
            "seconds": time.perf_counter() - t_start}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU (weak scaling)")
    ap.add_argument("--layers", type=int, default=N_LAYERS, help="extra measurements only: net depth (default 12)")
    ap.add_argument("--frame", type=int, default=FRAME, help="extra measurements only: samples per frame (default 16384)")
    ap.add_argument("--mode", choices=["train", "forward"], default="train",
                    help="train = the headline metric (BASELINE configs[2]/[3]); forward = eval-mode forward only "
                         "(configs[1], the enhancement.py path) - an extra measurement, same JSON shape")
    ap.add_argument("--torch-adam", action="store_true", help="use torch.optim.Adam instead of the fused HIP Adam (f1)")
    ap.add_argument("--gemm", choices=["split", "fp32", "bf16"], default=None,
                    help="GEMM arithmetic of the levels >= 16 samples: split = 3 x f16 MFMA on hi/lo fp16 halves of every fp32 "
                         "operand, fp32 accumulation (default, == WUNET_H3=1); fp32 = v_mfma_f32_16x16x4_f32 everywhere (WUNET_H3=0); "
                         "bf16 = bf16 operands, one MFMA pass (WUNET_H3=3; BASELINE configs[4], extra measurements only: outside "
                         "the 1e-4 fp32 parity bar)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the training step as ONE captured hipGraph (torch.cuda.CUDAGraph over forward + loss + backward + "
                         "fused Adam with its device-side step counter; SURVEY.md section 8 f2).  auto = on for one GPU with the fused "
                         "Adam: since the backward enqueues each layer's data gradient ahead of its side-stream weight gradient the "
                         "replay is as fast as the eager step (5.25 vs 5.26 ms back to back, profiles/r4_graph_vs_eager.txt) and a loop "
                         "that synchronises every step no longer pays the host's launch jitter (median 5.27 vs 5.5 ms); a failed "
                         "capture falls back to eager launches (reported in config.step_launch)")
    ap.add_argument("--native-rccl", action="store_true",
                    help="the gradient all-reduce through the library's own RCCL entry (include/wunet_hip.h wunet_comm_*: enqueued on the "
                         "backward's streams, capturable with --graph on) instead of torch.distributed's process group; at --gpus 1 the "
                         "bucketed collectives are issued anyway (world size 1), so their cost inside the step can be measured on one GPU")
    ap.add_argument("--seed", type=int, default=0, help="extra measurements only: seed of the initial weights and of the synthetic batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-fp32 and eval-forward side measurements")
    args = ap.parse_args()
    # the contract: rank 0 prints ONE JSON line.  Libraries write to the C-level stdout behind Python's back (RCCL prints a version
    # banner when a communicator is created): everything but the result line goes to stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if args.gemm is not None:
        os.environ["WUNET_H3"] = {"split": "1", "fp32": "0", "bf16": "3"}[args.gemm]
    h3_mode = os.environ.get("WUNET_H3", "1")
    split_gemm = h3_mode != "0"
    bf16_gemm = h3_mode in ("3", "4")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TEST HOOK, never a measurement: WUNET_BENCH_EMU=1 (tests/test_bench_contract.py) runs this file's multi-rank branches in the
    # GPU-less container - gloo, CPU tensors, the kernels on tests/emu's fiber emulator - so that `torchrun ... bench.py --gpus N`
    # stays correct by construction while no multi-GPU node is at hand.  The line it prints says so in `data`.
    emu = bool(os.environ.get("WUNET_BENCH_EMU"))
    if not emu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cpu") if emu else torch.device("cuda", local_rank)
    dsync = (lambda: None) if emu else torch.cuda.synchronize
    if not emu:
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)     # "nccl" is RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    pkg = importlib.import_module(PKG)
    parallel = importlib.import_module(PKG + ".parallel")
    engine_mod = importlib.import_module(PKG + ".engine")
    emu_engine = None
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_lib
        emu_engine = engine_mod.Engine(lib=importlib.import_module(PKG + "._lib").declare(emu_lib.lib()), host_memory=True)
        args.no_roofline = args.no_extras = args.no_cpu_baseline = True

    torch.manual_seed(args.seed)                 # same init on every rank (reference train.py:12)
    model = pkg.Model(n_layers=args.layers, channels_interval=CI).to(device).train()
    crit = pkg.smooth_l1_loss()
    optim_mod = importlib.import_module(PKG + ".optim")
    adam_cls = torch.optim.Adam if args.torch_adam else optim_mod.FusedAdam     # reference train.py:31-35
    opt = adam_cls(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
    if emu_engine is not None:
        model._engine_override = crit._engine_override = emu_engine
        if not args.torch_adam:
            opt._engine_override = emu_engine
    if world > 1 or args.native_rccl:
        # bucketed RCCL all-reduce inside the backward; the 1/world of the average is folded into the fused Adam step
        # (same rounding, four launches and 2 x 40 MB of traffic less per step) unless torch's optimiser is used
        comm = parallel.NativeComm() if args.native_rccl else None
        model.grad_sync = parallel.GradSync(n_buckets=4, scale_in_optimizer=not args.torch_adam, comm=comm, always_reduce=args.native_rccl)
        if not args.torch_adam:
            opt.grad_scale = 1.0 / world
    noisy, clean = synthetic_batch(args.batch, device, seed=rank + 1000 * args.seed, frame=args.frame)
    default_net = args.layers == N_LAYERS and args.frame == FRAME
    fwd_flop, fwd_bytes = net_flops_bytes(args.layers, CI, args.frame)
    step_flop = 3.0 * fwd_flop - 2.0 * (1 * CI * 15 * args.frame) if args.mode == "train" else fwd_flop   # no dgrad for encoder[0]
    step_bytes = 3.0 * fwd_bytes if args.mode == "train" else fwd_bytes

    if args.mode == "forward":
        model.eval()

    def step():
        if args.mode == "forward":
            with torch.no_grad():
                return model(noisy).sum()
        opt.zero_grad(set_to_none=True)
        out = model(noisy)
        loss = crit(clean, out)
        loss.backward()
        opt.step()
        return loss

    fused_adam = not args.torch_adam
    use_graph = not emu and args.mode == "train" and fused_adam and (
        (args.graph == "on" and (world == 1 or args.native_rccl)) or (args.graph == "auto" and world == 1))
    eager_step = step
    if use_graph:
        opt.device_step = True                   # the step counter and bias corrections live on the device: replayable
    for _ in range(args.warmup):
        step()
    if use_graph:
        # the same work as `step`, captured once: every kernel of forward, loss, backward (both streams) and the Adam step
        try:
            dsync()
            opt.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out_g = model(noisy)
                loss_g = crit(clean, out_g)
                loss_g.backward()
                opt.step()
            opt.advance_host_step(-1)                # capture does not execute

            def step():
                graph.replay()
                opt.advance_host_step(1)
                return loss_g
            for _ in range(2):
                step()
        except Exception as e:                       # noqa: BLE001 - `auto` must never cost the bench line: eager launches instead
            if args.graph == "on":
                raise
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            use_graph = False
            step = eager_step
            dsync()
    if world > 1:
        dist.barrier()
    dsync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    enqueue = time.perf_counter() - t0          # host time to issue the K steps (the GPU runs behind it)
    dsync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    if os.environ.get("WUNET_STAMP") and not emu and args.mode == "train":
        # measurement hook (csrc/wunet_backward.cpp, WUNET_STAMP): when the backward's two chains ran inside the LAST step above - device wall clock,
        # 100 MHz - written to stderr, layer by layer in backward order: start of the data gradient, start of the weight gradient, relative to the first stamp
        st_lib = ctypes.CDLL(importlib.import_module(PKG + "._lib").LIB_PATH)
        buf = (ctypes.c_ulonglong * 128)()
        if st_lib.wunet_debug_stamps(buf, 128) == 0:
            t_first = min(v for v in buf if v)
            nl = 2 * args.layers + 1
            print("[stamps] layer: data-gradient chain reaches it at / weight gradient starts at (us after the first stamp), step_launch = %s" % ("graph" if use_graph else "eager"), file=sys.stderr)
            for i in range(nl - 1, -1, -1):
                a, b = buf[2 * i], buf[2 * i + 1]
                print("[stamps] layer %2d  main %8.1f  side %8.1f  lag %8.1f" % (i, (a - t_first) / 100.0 if a else -1, (b - t_first) / 100.0 if b else -1, (b - a) / 100.0 if a and b else -1), file=sys.stderr)
            print("[stamps] join: main chain done %8.1f, side chain done %8.1f" % ((buf[126] - t_first) / 100.0, (buf[127] - t_first) / 100.0), file=sys.stderr)
    # a second pass of the same K steps timed one by one with device events: the median is robust against a clock ramp or a
    # hiccup inside the short timed region above (which stays the headline, as the contract defines it)
    step_ms = []
    for _ in range(0 if (os.environ.get("WUNET_BENCH_NO_MEDIAN") or emu) else args.steps):      # (switch: the PMC passes count launches per step)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        e1.synchronize()
        step_ms.append(e0.elapsed_time(e1))
    step_ms.sort()
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None

    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = args.batch * world * args.steps / elapsed

    # the gradient exchange of this job, per rank: what part of the bucketed all-reduces did not hide under the backward (device events
    # around the join of a few extra steps), the buckets, the world size the transport reports
    exchange = None
    sync = getattr(model, "grad_sync", None)
    if sync is not None and args.mode == "train":
        sync.measure = True
        for _ in range(1 if emu else 5):
            eager_step()
        dsync()
        sync.measure = False
        ex = sync.exposed_ms()
        mine = torch.tensor([sum(ex) / max(len(ex), 1)], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)] if world > 1 else [mine]
        if world > 1:
            dist.all_gather(allr, mine)
        exchange = sync.describe(model._wunet_params(), 2 * args.layers + 1)
        exchange["allreduce_exposed_ms_per_rank"] = [round(float(t.item()), 4) for t in allr]
        exchange["measured_over_steps"] = len(ex)
        # what every rank actually ran (a scaling record must explain itself: eager or one graph per step, which transport, the world size
        # that transport reports, the device)
        mine_desc = {"rank": rank, "step_launch": "one hipGraph replay per step" if use_graph else "eager launches",
                     "transport": exchange["transport"], "world_seen": exchange["world"], "device": str(device)}
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, mine_desc)
        else:
            per_rank = [mine_desc]
        exchange["per_rank"] = per_rank

    roofline = None
    nprof = max(1, min(args.steps, 5))
    if rank != 0 and not args.no_roofline:
        for _ in range(nprof):          # keep the collectives of the profiled pass matched on every rank
            eager_step()
    if rank == 0 and not args.no_roofline:
        # per-kernel durations over a second pass of the same steps (the per-kernel events need eager launches: a captured graph
        # cannot be instrumented)
        lib = engine_mod.default_engine().lib
        rows = kernel_rows(lib, eager_step, nprof)
        deep16 = args.layers == 16 and args.frame == 65536 and args.batch == 32
        section = ("train" if default_net and args.batch == 64 and split_gemm and not bf16_gemm else
                   "gemm_fp32" if default_net and args.batch == 64 and not split_gemm else
                   "deep16_bf16" if deep16 and bf16_gemm else "-") if args.mode == "train" else "-"
        roofline = roofline_of(rows, nprof, load_pmc_traffic(None if section == "train" else section), args.batch * step_bytes)

    # ---- extras the driver should see next to the headline (same box, same process): the exact-fp32 arithmetic and the
    #      eval-mode forward (BASELINE configs[1])
    extras = None
    if rank == 0 and world == 1 and args.mode == "train" and default_net and not args.no_extras:
        extras = {}
        # the contract's K = 20 steps are a 0.1 s window; 200 back-to-back steps and the median of 200 event-timed steps of the SAME step
        # function resolve a 1 - 2 % change (the boxes of the pool differ by more than that: compare within one run)
        dsync()
        t_l = time.perf_counter()
        for _ in range(200):
            step()
        dsync()
        long_ms = (time.perf_counter() - t_l) / 200 * 1e3
        per = []
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); step(); e1.record(); e1.synchronize()
            per.append(e0.elapsed_time(e1))
        per.sort()
        extras["steps200"] = {"what": "the headline step, 200 back-to-back steps / median and quartiles of 200 steps timed one by one with device events",
                              "ms_per_step": long_ms, "frames_per_s": args.batch / long_ms * 1e3,
                              "ms_per_step_median": per[100], "ms_per_step_p25": per[50], "ms_per_step_p75": per[150]}
        # (first, while the allocator still holds the headline run's blocks: every forward takes its workspace from torch)
        model.eval()

        def fwd():
            with torch.no_grad():
                model(noisy)
        extras["eval_forward"] = timed(fwd, 10, 50, args.batch, "eval-mode forward only (BASELINE configs[1], enhancement.py path), default GEMM arithmetic",
                                       median_steps=50)
        if not args.no_roofline:
            # BASELINE configs[1] carries its own roofline: the dominant GEMM kernel of the eval forward (HIP events of the same calls)
            # and the PMC traffic of its own passes (tools/measure_round.sh: `bench.py --mode forward` under --pmc)
            extras["eval_forward"]["roofline"] = roofline_of(kernel_rows(engine_mod.default_engine().lib, fwd, 3), 3, load_pmc_traffic("eval_forward"),
                                                             args.batch * fwd_bytes)
            extras["eval_forward"]["whole_forward_tflops"] = extras["eval_forward"]["frames_per_s"] * fwd_flop / 1e12
        model.train()
        torch.manual_seed(0)
        m32 = pkg.Model(n_layers=args.layers, channels_interval=CI).to(device).train()
        m32._engine_override = engine_mod.Engine(h3=0)
        c32 = pkg.smooth_l1_loss()
        c32._engine_override = m32._engine_override
        o32 = adam_cls(m32.parameters(), lr=1e-3, betas=(0.9, 0.999))

        def step32():
            o32.zero_grad(set_to_none=True)
            c32(clean, m32(noisy)).backward()
            o32.step()
        extras["gemm_fp32"] = timed(step32, 3, 10, args.batch, "training step, every GEMM on v_mfma_f32_16x16x4_f32 (exact fp32: WUNET_H3=0)")
        extras["gemm_fp32"]["dtype"] = "f32"
        if not args.no_roofline:
            # the strictly-fp32 figure carries its own roofline: dominant v_mfma_f32_16x16x4_f32 kernel against the 157.3 TF fp32 peak
            extras["gemm_fp32"]["roofline"] = roofline_of(kernel_rows(m32._engine_override.lib, step32, 3), 3, load_pmc_traffic("gemm_fp32"),
                                                          args.batch * step_bytes)
        del m32, o32, c32
        # BASELINE.json configs[4]: "24-level / 65536-sample deep variant, bf16, batch=32" - 16 levels is the deepest net that exists
        # at 65536 samples: bf16 operands, one bf16 MFMA pass, f32 accumulation / BatchNorm / gradients
        try:
            torch.manual_seed(0)
            e16 = engine_mod.Engine(h3=3)
            m16 = pkg.Model(n_layers=16, channels_interval=CI).to(device).train()
            m16._engine_override = e16
            c16 = pkg.smooth_l1_loss()
            c16._engine_override = e16
            o16 = adam_cls(m16.parameters(), lr=1e-3, betas=(0.9, 0.999))
            n16, cl16 = synthetic_batch(32, device, seed=rank, frame=65536)

            def step16():
                o16.zero_grad(set_to_none=True)
                c16(cl16, m16(n16)).backward()
                o16.step()
            d16 = timed(step16, 10, 50, 32, "training step of the 16-level / 65536-sample net at batch 32, bf16 GEMM operands (WUNET_H3=3): BASELINE.json "
                                            "configs[4] (24 levels do not exist at 65536 samples, SURVEY.md section 0); outside the 1e-4 fp32 parity bar, "
                                            "checked against the reference under bf16 autocast (tests/test_gpu_parity.py)", median_steps=50)
            d16["value"], d16["unit"] = d16["frames_per_s"], "65536-sample frames/s"
            d16["dtype"] = "bf16 operands, f32 accumulate / BatchNorm / gradients"
            f16fl, f16by = net_flops_bytes(16, CI, 65536)
            d16["whole_step_tflops"] = d16["frames_per_s"] * (3.0 * f16fl - 2.0 * CI * 15 * 65536) / 1e12
            if not args.no_roofline:
                d16["roofline"] = roofline_of(kernel_rows(e16.lib, step16, 3), 3, load_pmc_traffic("deep16_bf16"), 32 * 3.0 * f16by)
            extras["deep16_bf16"] = d16
            del m16, o16, c16, n16, cl16
        except Exception as e:      # (an extra must never take the headline line down with it)
            extras["deep16_bf16"] = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        per_gpu_fps = frames_per_s / world
        result = {
            "metric": ("16384-sample frames/sec fwd+bwd, 12-level Wave-U-Net" if args.mode == "train"
                       else "16384-sample frames/sec eval forward only, 12-level Wave-U-Net (extra, BASELINE configs[1])")
                      if default_net else f"{args.frame}-sample frames/sec, {args.layers}-level Wave-U-Net, mode={args.mode} (extra)",
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_median": median_ms,
            "value_at_median": (args.batch * world / (median_ms * 1e-3)) if median_ms else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 operands, f32 accumulate / BatchNorm / gradients (levels >= 16 samples: 1 x bf16 MFMA; the rest f32 MFMA)" if bf16_gemm
                      else "f32 (levels >= 16 samples: 3 x f16 MFMA on hi/lo fp16 halves, fp32 accumulate; the rest f32 MFMA)" if split_gemm
                      else "f32"),
            "data": "synthetic" if not emu else "synthetic; CPU EMULATOR RUN (WUNET_BENCH_EMU test hook) - NOT A MEASUREMENT",
            "config": {"workload": f"unet_basic {args.layers}-level, {args.frame}-sample frames, batch={args.batch} per GPU, fp32, "
                                   "training-mode forward + smooth_l1 + backward + " + ("torch.optim.Adam" if args.torch_adam else "fused HIP Adam") + " step "
                                   "(BASELINE.json configs[2]; configs[3] when n_gpus>1)",
                       "global_batch": args.batch * world, "frame": args.frame,
                       "parallelism": f"dp{world}" + (" (RCCL bucketed all-reduce, per-GPU BatchNorm)" if world > 1 else "")
                                      + (" [collectives through the library's RCCL entry]" if args.native_rccl else "")},
            "whole_step_tflops_per_gpu": per_gpu_fps * step_flop / 1e12,
            # against the roof of the arithmetic that ran: 2500 / 3 TF for the fp16-split GEMMs, 2500 TF bf16, 157.3 TF exact fp32
            "whole_step_frac_of_gemm_peak": per_gpu_fps * step_flop / 1e12 / (PEAK_F16_MFMA_TFLOPS if bf16_gemm else PEAK_SPLIT_TFLOPS if split_gemm
                                                                                else PEAK_FP32_MFMA_TFLOPS),
            "gemm": "bf16" if bf16_gemm else "split" if split_gemm else "fp32",
            "whole_step_algorithmic_hbm_frac": per_gpu_fps * step_bytes / 1e9 / PEAK_HBM_GBS,
            "final_loss": final_loss,
            "host_enqueue_ms_per_step": enqueue / args.steps * 1e3,
            "step_launch": "one hipGraph replay per step" if use_graph else "eager launches",
            "gradient_exchange": exchange,
            "roofline": roofline, "cpu_baseline": cpu, "extras": extras,
        }
        # the strict-fp32 training step and the eval forward (BASELINE configs[1]) beside the headline, not only inside `extras`
        if extras:
            g32, evf = extras.get("gemm_fp32") or {}, extras.get("eval_forward") or {}
            result["gemm_fp32"] = {"value": g32.get("frames_per_s"), "unit": "frames/s", "ms_per_step": g32.get("ms_per_step"), "dtype": "f32",
                                   "roofline_frac": (g32.get("roofline") or {}).get("frac"), "roofline_bound": (g32.get("roofline") or {}).get("bound")}
            result["eval_forward"] = {"value": evf.get("frames_per_s"), "unit": "frames/s", "ms_per_step": evf.get("ms_per_step"),
                                      "ms_per_step_median": evf.get("ms_per_step_median"),
                                      "roofline_frac": (evf.get("roofline") or {}).get("frac"), "roofline_bound": (evf.get("roofline") or {}).get("bound")}
        result_out.write(json.dumps(result) + "\n")
        result_out.flush()


if __name__ == "__main__":
    main()
