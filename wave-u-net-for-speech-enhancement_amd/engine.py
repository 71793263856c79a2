"""Python face of the C ABI: owns per-shape contexts, turns torch tensors into raw device pointers
and enqueues the HIP path on torch's current stream.  PyTorch is plumbing here (device memory,
streams); every arithmetic kernel lives in csrc/."""
import collections
import ctypes
import os
import threading
import weakref

import torch

from . import _lib

LOSS_KINDS = {"mse": 0, "l1": 1, "smooth_l1": 2}


class WunetError(RuntimeError):
    pass


def bump_versions(tensors):
    """Tell autograd's version counters that a kernel wrote these tensors through raw pointers (what every in-place torch op does
    by itself).  torch >= 2.1 has the call (a list where it is accepted, else tensor by tensor); older ones are left as they were."""
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is None:
        return
    try:
        inc(tensors)                     # (recent torch takes the whole list)
    except TypeError:                    # torch 2.1 - 2.4: one tensor per call
        for t in tensors:
            inc(t)


class _ParamSignature:
    """What the eval-mode weight-pack cache is valid for: THESE parameter tensor objects (weak references - an address the caching
    allocator hands to the next model's weights is not the same tensor), at these addresses and autograd versions, and no device-side
    rewrite (Engine.weights_generation) since."""
    __slots__ = ("refs", "state", "gen")

    def __init__(self, params, gen):
        self.refs = [weakref.ref(p) for p in params]
        self.state = tuple((p.data_ptr(), p._version) for p in params)
        self.gen = gen

    def matches(self, other):
        """self: the cached entry; other: the signature of the call's parameters (alive by construction)."""
        if self.gen != other.gen or self.state != other.state or len(self.refs) != len(other.refs):
            return False
        return all(a() is b() and a() is not None for a, b in zip(self.refs, other.refs))


class FlatGrads:
    """The gradient tensors of `Engine.backward` as ONE flat fp32 buffer + the element offset of every parameter in it."""

    def __init__(self, flat, offsets):
        self.flat, self.offsets = flat, offsets


def _load_torch_ext():
    """The optional C++ marshalling layer (torch_ext/wunet_torch.cpp, built by __graft_entry__.build()); None when it is not there or
    when WUNET_LIB_PATH selects another build of the library (the extension is linked to csrc/libwunet_hip.so)."""
    if os.environ.get("WUNET_LIB_PATH") or os.environ.get("WUNET_NO_TORCH_EXT"):
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "torch_ext", "_wunet_torch.so")
    stamp = path + ".torch"              # the torch version the extension was built against (__graft_entry__._build_torch_ext)
    if not os.path.exists(path):
        return None
    try:
        if open(stamp).read().strip() != torch.__version__:
            return None                  # built against another torch: its ABI is not this one's - the ctypes path instead
    except OSError:
        return None
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_wunet_torch", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except Exception:                    # noqa: BLE001 - optional: a stale build (other torch) falls back to ctypes
        return None


class Engine:
    """One per process is enough (see `default_engine`).  `lib`/`host_memory` exist so the CPU test
    suite can drive the same host logic against the emulator build with CPU tensors; product code
    never passes them."""

    def __init__(self, lib=None, host_memory=False, h3=None):
        self.lib = lib if lib is not None else _lib.load_hip()
        self.host_memory = host_memory
        # GEMM arithmetic of the levels >= 16 samples (include/wunet_hip.h: wunet_set_h3): 1 = fp16-split MFMA where the grid
        # fills the chip (default), 0 = fp32 MFMA everywhere (WUNET_H3=0), 2 = fp16-split wherever it can run (tests),
        # 3 = bf16 operands, one MFMA pass, on the planner's layers (BASELINE configs[4]), 4 = bf16 wherever it can run (tests)
        self.h3 = int(os.environ.get("WUNET_H3", "1")) if h3 is None else int(h3)
        # contexts are keyed by shape AND device (the ctx owns a weight-gradient side stream per device it runs on; replicas
        # of one shape on several devices - a stock nn.DataParallel wrap, trainer/base_trainer.py:26-27 - get their own),
        # least-recently-used ones are destroyed beyond MAX_CONTEXTS (variable-length inference would otherwise leak one per shape)
        self._ctx = collections.OrderedDict()
        self._lock = threading.Lock()
        self._eval_ws = collections.OrderedDict()     # (shape, device, stream) -> (workspace of the last eval forward, parameter signature)
        # Bumped by everything that rewrites weights WITHOUT moving a Python version counter: the replay of a captured step graph
        # (FusedAdam.advance_host_step) - part of the eval cache's signature, so an eval forward after N replays re-packs.
        self.weights_generation = 0
        # per-step calls marshalled in C++ when the extension is built and this engine drives the HIP library it is linked to
        self._fast = _load_torch_ext() if (lib is None and not host_memory) else None

    # ------------------------------------------------------------------ helpers
    def _check(self, rc):
        if rc != 0:
            raise WunetError(f"wunet error {rc}: {self.lib.wunet_last_error().decode()}")

    MAX_CONTEXTS = 32

    def _ctx_for(self, n_layers, ci, batch, length, device=None):
        """The context of one (shape, device), created on first use.  Every C call that takes the handle runs inside
        `with self._using(...) as h` (below): a handle that some thread still holds is never destroyed by the eviction of
        another thread (stock nn.DataParallel calls forward from one thread per replica, trainer/base_trainer.py:26-27)."""
        key = (n_layers, ci, batch, length, str(device))
        with self._lock:
            ent = self._ctx.get(key)
            if ent is None:
                h = ctypes.c_void_p()
                self._check(self.lib.wunet_create(n_layers, ci, batch, length, ctypes.byref(h)))
                if self.h3:
                    self._check(self.lib.wunet_set_h3(h, self.h3))
                ent = self._ctx[key] = [h, 0]                    # [handle, holders]
            else:
                self._ctx.move_to_end(key)
            ent[1] += 1
            # least-recently-used contexts nobody holds go beyond MAX_CONTEXTS (oldest first; the one just handed out is held)
            for k in [k for k, e in self._ctx.items() if e[1] == 0][:max(0, len(self._ctx) - self.MAX_CONTEXTS)]:
                self.lib.wunet_destroy(self._ctx.pop(k)[0])       # (HIP defers the side stream's destruction until its work has drained)
        return key, ent

    def _release(self, ent):
        with self._lock:
            ent[1] -= 1

    def _using(self, n_layers, ci, batch, length, device=None):
        return _CtxHold(self, *self._ctx_for(n_layers, ci, batch, length, device))

    def _require(self, t, name):
        if t.dtype != torch.float32 and t.dtype != torch.int64:
            raise WunetError(f"{name}: expected float32, got {t.dtype}")
        if not t.is_contiguous():
            raise WunetError(f"{name}: tensor must be contiguous")
        if self.host_memory:
            if t.device.type != "cpu":
                raise WunetError(f"{name}: emulator engine takes CPU tensors")
        elif t.device.type != "cuda":
            raise WunetError(f"{name}: the HIP path needs tensors on an MI355X (got device {t.device}); "
                             "there is no CPU fallback")

    def _stream(self, device):
        if self.host_memory:
            return None
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    @staticmethod
    def _ptrs(tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def _require_all(self, tensors, what, device=None):
        """`_require` of a whole list (+ same device as `device`) in one pass - the 102 parameters, 75 buffers and 4 x 102 Adam
        tensors of a step: the per-tensor form cost the host ~0.8 us per tensor, a third of the time it needs to issue a step.
        The slow path below names the offender."""
        f32, i64 = torch.float32, torch.int64
        on_gpu = not self.host_memory
        idx = -1 if device is None or device.index is None else device.index
        for t in tensors:
            dt = t.dtype
            if (dt is not f32 and dt is not i64) or not t.is_contiguous() or t.is_cuda is not on_gpu or (on_gpu and idx >= 0 and t.get_device() != idx):
                break
        else:
            return
        for k, t in enumerate(tensors):
            self._require(t, f"{what}[{k}]")
            if device is not None and t.device != device:
                raise WunetError(f"{what}[{k}] lives on {t.device}, the call runs on {device}: parameters and input live on different devices")

    def _device_guard(self, device):
        return torch.cuda.device(device) if not self.host_memory else _NullCtx()

    # ------------------------------------------------------------------ network
    def forward(self, n_layers, ci, noisy, params, running, nbt, training, with_backward):
        """Returns (enhanced, workspace).  Keep `workspace` alive until backward has been enqueued."""
        if noisy.dim() != 3 or noisy.shape[1] != 1:
            raise WunetError(f"input must be [batch, 1, samples], got {tuple(noisy.shape)}")
        # eval mode (enhancement.py:57-69: the same weights over every chunk of every file): the workspace of the previous eval forward
        # on this (shape, device, stream) is used again, and while no parameter has changed - addresses and autograd version counters,
        # which every in-place torch op, load_state_dict and FusedAdam.step move - the weight packs in it are not rebuilt
        cache = sig = None
        if training and self._eval_ws:
            self.drop_eval_cache()           # training resumes: the eval workspaces (up to MAX_EVAL_WORKSPACES whole forwards) go back to the allocator
        if (not training and not with_backward and not self.host_memory and noisy.is_cuda and noisy.dtype == torch.float32
                and not os.environ.get("WUNET_NO_EVAL_CACHE")):                       # (switch: A/B; a CPU / non-float input falls through to _require's error)
            key = (n_layers, ci, noisy.shape[0], noisy.shape[2], str(noisy.device), torch.cuda.current_stream(noisy.device).cuda_stream)
            try:
                sig = _ParamSignature(params, self.weights_generation)
            except (RuntimeError, TypeError):                                         # inference tensors have no version counter: no cache
                sig = None
            if sig is not None:
                with self._lock:
                    cache = self._eval_ws.pop(key, None)         # (popped: a second thread on the same key takes a fresh workspace)
                if cache is not None and not cache[1].matches(sig):
                    cache = (cache[0], None)
        if self._fast is not None:
            with self._using(n_layers, ci, noisy.shape[0], noisy.shape[2], noisy.device) as h:
                try:
                    out, ws = self._fast.forward(h.value, noisy, params, running, nbt, bool(training), bool(with_backward),
                                                 cache[0] if cache else None, bool(cache and cache[1] is not None))
                except RuntimeError as e:
                    raise WunetError(str(e).split("\n")[0]) from None
            if sig is not None:
                self._remember_eval_ws(key, ws, sig)
            return out, ws
        self._require(noisy, "input")
        self._require_all(params, "param", noisy.device)
        self._require_all(running, "buffer", noisy.device)
        self._require_all(nbt, "buffer", noisy.device)
        B, _, T = noisy.shape
        with self._using(n_layers, ci, B, T, noisy.device) as h, self._device_guard(noisy.device):
            nbytes = self.lib.wunet_workspace_bytes(h, 1 if with_backward else 0)
            reuse = cache is not None and cache[0].numel() * 4 >= nbytes
            ws = cache[0] if reuse else torch.empty(nbytes // 4, dtype=torch.float32, device=noisy.device)
            out = torch.empty_like(noisy)
            flags = (1 if with_backward else 0) | (2 if reuse and cache[1] is not None else 0)          # WUNET_FWD_SAVE | WUNET_FWD_PACKS_VALID
            self._check(self.lib.wunet_forward(h, noisy.data_ptr(), self._ptrs(params), self._ptrs(running),
                                               self._ptrs(nbt), 1 if training else 0, flags, ws.data_ptr(),
                                               out.data_ptr(), self._stream(noisy.device)))
        if sig is not None:
            self._remember_eval_ws(key, ws, sig)
        return out, ws

    MAX_EVAL_WORKSPACES = 4

    def _remember_eval_ws(self, key, ws, sig):
        with self._lock:
            self._eval_ws[key] = (ws, sig)
            while len(self._eval_ws) > self.MAX_EVAL_WORKSPACES:            # (variable-length inference: the oldest shapes go)
                self._eval_ws.pop(next(iter(self._eval_ws)))

    def note_weights_changed(self):
        """Weights were rewritten on the device without a Python-side in-place op (a graph replay that contains the optimiser step):
        cached eval weight packs of every model on this engine are stale from here on."""
        self.weights_generation += 1

    def drop_eval_cache(self):
        """Forget the eval-mode workspaces (and with them the cached weight packs): for a caller that changed weights behind autograd's
        back (`p.data[...] = ...` does not move a version counter)."""
        with self._lock:
            self._eval_ws.clear()

    def backward(self, n_layers, ci, noisy, params, enhanced, grad_enhanced, ws, grads, layer_range=None, join=True):
        """join=False: the caller's stream does not wait for the range's weight gradients (side stream); make a stream see them
        with `join_weight_gradients` (parallel.GradSync does, on the stream that launches the bucket's all-reduce)."""
        B, _, T = noisy.shape
        self._require(grad_enhanced, "grad_output")
        nl = 2 * n_layers + 1
        lb, le = layer_range if layer_range is not None else (0, nl)
        if self._fast is not None and isinstance(grads, FlatGrads):
            with self._using(n_layers, ci, B, T, noisy.device) as h:
                try:
                    self._fast.backward_range(h.value, noisy, params, enhanced, grad_enhanced, ws, grads.flat, grads.offsets, lb, le, bool(join))
                except RuntimeError as e:
                    raise WunetError(str(e).split("\n")[0]) from None
            return
        fn = self.lib.wunet_backward_range if join else self.lib.wunet_backward_range_async
        if isinstance(grads, FlatGrads):         # one flat buffer + element offsets: the pointers are arithmetic, no 102 views first
            base = grads.flat.data_ptr()
            gptrs = (ctypes.c_void_p * len(grads.offsets))(*[base + 4 * o for o in grads.offsets])
        else:
            gptrs = self._ptrs(grads)
        with self._using(n_layers, ci, B, T, noisy.device) as h, self._device_guard(noisy.device):
            self._check(fn(h, noisy.data_ptr(), self._ptrs(params), enhanced.data_ptr(), grad_enhanced.data_ptr(), ws.data_ptr(),
                           gptrs, lb, le, self._stream(noisy.device)))

    def join_weight_gradients(self, n_layers, ci, noisy):
        """Torch's current stream waits for everything the weight-gradient side stream has been given so far."""
        B, _, T = noisy.shape
        with self._using(n_layers, ci, B, T, noisy.device) as h, self._device_guard(noisy.device):
            self._check(self.lib.wunet_backward_join(h, self._stream(noisy.device)))

    def layer_output(self, n_layers, ci, batch, length, ws, layer):
        """Raw conv output (pre-BatchNorm) of conv layer `layer` as a view into a workspace (tests/profiling)."""
        off, ch, ln = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int()
        with self._using(n_layers, ci, batch, length, ws.device) as h:
            self._check(self.lib.wunet_layer_info(h, layer, ctypes.byref(off), ctypes.byref(ch), ctypes.byref(ln)))
        return ws[off.value: off.value + batch * ch.value * ln.value].view(batch, ch.value, ln.value)

    # ------------------------------------------------------------------ losses
    def loss_forward(self, kind, clean, enhanced):
        self._require(clean, "clean")
        self._require(enhanced, "enhanced")
        if clean.shape != enhanced.shape:
            raise WunetError(f"loss: shape mismatch {tuple(clean.shape)} vs {tuple(enhanced.shape)}")
        out = torch.empty((), dtype=torch.float32, device=enhanced.device)
        scratch = torch.empty(self.lib.wunet_loss_scratch_bytes() // 8, dtype=torch.float64, device=enhanced.device)
        with self._device_guard(enhanced.device):
            self._check(self.lib.wunet_loss_forward(LOSS_KINDS[kind], clean.data_ptr(), enhanced.data_ptr(),
                                                    enhanced.numel(), out.data_ptr(), scratch.data_ptr(),
                                                    self._stream(enhanced.device)))
        return out

    def loss_backward(self, kind, clean, enhanced, grad_loss):
        g = torch.empty_like(enhanced)
        gl = grad_loss.to(torch.float32).contiguous()
        with self._device_guard(enhanced.device):
            self._check(self.lib.wunet_loss_backward(LOSS_KINDS[kind], clean.data_ptr(), enhanced.data_ptr(),
                                                     gl.data_ptr(), enhanced.numel(), g.data_ptr(),
                                                     self._stream(enhanced.device)))
        return g


def _adam_step(self, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0, step_dev=None, hyper_dev=None):
    """One fused Adam step over a list of tensors.  step_dev (int64 device scalar) + hyper_dev (2 floats):
    the step count lives on the device and the call increments it (capturable in a hipGraph); otherwise `step` is the host's."""
    if self._fast is not None:
        try:
            self._fast.adam_step(params, grads, exp_avg, exp_avg_sq, float(lr), float(beta1), float(beta2), float(eps), int(step),
                                 float(grad_scale), step_dev, hyper_dev)
        except RuntimeError as e:
            raise WunetError(str(e).split("\n")[0]) from None
        return
    for name, ts in (("param", params), ("grad", grads), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        self._require_all(ts, "adam " + name, params[0].device)
    n = len(params)
    numels = (ctypes.c_size_t * n)(*[p.numel() for p in params])
    dev = params[0].device
    with self._device_guard(dev):
        self._check(self.lib.wunet_adam_step(n, self._ptrs(params), self._ptrs(grads), self._ptrs(exp_avg),
                                             self._ptrs(exp_avg_sq), numels, float(lr), float(beta1), float(beta2),
                                             float(eps), int(step), float(grad_scale),
                                             step_dev.data_ptr() if step_dev is not None else None,
                                             hyper_dev.data_ptr() if hyper_dev is not None else None, self._stream(dev)))


Engine.adam_step = _adam_step


def _crop_windows(self, mixture_flat, clean_flat, starts, mixture, clean):
    """mixture[b, 0, :] = mixture_flat[starts[b] : starts[b] + L], clean likewise (one launch for the batch)."""
    for t in (mixture_flat, clean_flat, starts, mixture, clean):
        self._require(t, "crop tensor")
    if starts.dtype != torch.int64 or mixture_flat.dtype != torch.float32 or mixture.dtype != torch.float32:
        raise WunetError("crop_windows: float32 arrays and int64 starts expected")
    if mixture_flat.numel() != clean_flat.numel() or mixture.shape != clean.shape or starts.numel() != mixture.shape[0]:
        raise WunetError("crop_windows: shape mismatch")
    B, L = mixture.shape[0], mixture.shape[-1]
    with self._device_guard(mixture.device):
        self._check(self.lib.wunet_crop_windows(mixture_flat.data_ptr(), clean_flat.data_ptr(), starts.data_ptr(), mixture_flat.numel(),
                                                B, L, mixture.data_ptr(), clean.data_ptr(), self._stream(mixture.device)))


Engine.crop_windows = _crop_windows


class _CtxHold:
    """`with engine._using(shape...) as handle`: the context cannot be evicted while the block runs."""

    def __init__(self, engine, key, ent):
        self.engine, self.key, self.ent = engine, key, ent

    def __enter__(self):
        return self.ent[0]

    def __exit__(self, *a):
        self.engine._release(self.ent)
        return False


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_DEFAULT = None
_DEFAULT_LOCK = threading.Lock()


def default_engine():
    global _DEFAULT
    with _DEFAULT_LOCK:
        if _DEFAULT is None:
            _DEFAULT = Engine()
    return _DEFAULT
