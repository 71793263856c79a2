"""Data parallelism for the hot path: one process per GPU, RCCL all-reduce of the flat gradient
buffer over xGMI, bucketed in backward-completion order and overlapped with the remaining backward
kernels.  Replaces the reference's single-process torch.nn.DataParallel wrap
(/root/reference/trainer/base_trainer.py:26-27).

Semantics kept from DataParallel: BatchNorm statistics stay local to each replica (no SyncBN),
running statistics are not reduced, and with equal shards the summed gradient of the global-mean
loss equals the average of the per-shard mean-loss gradients - so buckets are summed and scaled by
1/world_size.  Every rank then takes the identical optimiser step; no parameter broadcast follows.
"""
import ctypes

import torch
import torch.distributed as dist


class NativeComm:
    """The RCCL communicator behind the C ABI (include/wunet_hip.h: wunet_comm_*): one per process and GPU.  torch.distributed is
    used ONLY to carry rank 0's 128-byte RCCL id to the other ranks (any initialised backend does; pass `comm_id` to do without).
    `all_reduce_` enqueues the in-place sum on torch's CURRENT stream like a kernel - it can be captured into the step's hipGraph,
    which torch's own process-group collectives (their watchdog, their private stream) make awkward."""

    def __init__(self, engine=None, group=None, comm_id=None, world=None, rank=None):
        from .engine import default_engine
        self.engine = engine if engine is not None else default_engine()
        lib = self.engine.lib
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        elif rank is None:
            raise ValueError("NativeComm: pass `rank` together with `world`")
        self.world, self.rank = world, rank
        buf = (ctypes.c_ubyte * 128)()
        if comm_id is None:
            box = [None]
            if rank == 0:
                rc = lib.wunet_comm_unique_id(buf)
                if rc != 0 and world > 1:           # (world size 1 runs without RCCL: the CPU test build)
                    self.engine._check(rc)
                box[0] = bytes(buf)
            if world > 1:
                # (src is a GLOBAL rank: the group's rank 0 drew the id)
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast_object_list(box, src=src, group=group)
            comm_id = box[0]
        buf = (ctypes.c_ubyte * 128)(*comm_id)
        self.handle = ctypes.c_void_p()
        self.engine._check(lib.wunet_comm_create(buf, world, rank, ctypes.byref(self.handle)))

    def all_reduce_(self, tensor):
        if tensor.dtype != torch.float32 or not tensor.is_contiguous():
            raise TypeError("NativeComm.all_reduce_: contiguous float32 expected")
        stream = None if self.engine.host_memory else ctypes.c_void_p(torch.cuda.current_stream(tensor.device).cuda_stream)
        self.engine._check(self.engine.lib.wunet_comm_allreduce_sum(self.handle, tensor.data_ptr(), tensor.numel(), stream))
        return tensor

    def close(self):
        if self.handle:
            self.engine.lib.wunet_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


def bucket_ranges(param_numels, n_conv_layers, n_buckets):
    """Split conv layers [0, NL) into <= n_buckets contiguous ranges, walked in backward order
    (last layer first), balanced by gradient bytes.  Layer i owns params 4i..4i+3; the output head's
    two tensors belong to the last layer's bucket.  Returns [(layer_begin, layer_end, flat_begin, flat_end)]
    in execution order."""
    nl = n_conv_layers
    per_layer = [sum(param_numels[4 * i:4 * i + 4]) for i in range(nl)]
    per_layer[nl - 1] += sum(param_numels[4 * nl:4 * nl + 2])
    offsets = [0]
    for v in param_numels:
        offsets.append(offsets[-1] + v)
    total = sum(per_layer)
    target = total / max(1, n_buckets)
    ranges, end, acc = [], nl, 0
    for i in range(nl - 1, -1, -1):
        acc += per_layer[i]
        remaining_buckets = n_buckets - len(ranges) - 1
        if (acc >= target and remaining_buckets > 0 and i > 0) or i == 0:
            flat_end = offsets[4 * end] if end < nl else offsets[-1]
            ranges.append((i, end, offsets[4 * i], flat_end))
            end, acc = i, 0
    return ranges


class GradSync:
    """Attach with `model.grad_sync = GradSync(...)`; the model's backward then runs the layer
    ranges through wunet_backward_range_async and enqueues one asynchronous all-reduce per finished bucket.

    Stream structure on the GPU (one process per GPU): the backward's data-gradient chain runs on torch's current stream S, the
    weight gradients on the library's side stream W.  Only the collective needs a bucket's weight gradients, so S is never made
    to wait for W at a bucket boundary: a launch stream C waits for S (event) and for W (wunet_backward_join) and the
    all-reduce is enqueued from C; RCCL's own stream then waits for C.  S waits for the collectives once, after the last range.

    scale_in_optimizer=True: the 1/world_size of the average is left to the optimiser step (optim.FusedAdam.grad_scale - the
    same rounding, four launches and 2 x 40 MB of traffic less); the default scales the buckets here so any optimiser works.
    With it, `param.grad` and `model.last_flat_grad` hold the SUM over the ranks until the step: anything else that reads them
    (gradient clipping, logging a norm) must apply the 1/world itself - or use the default."""

    def __init__(self, process_group=None, n_buckets=4, always_reduce=False, scale_in_optimizer=False, comm=None):
        self.group = process_group
        self.comm = comm                        # a NativeComm: the collectives go through the library's own RCCL entry (capturable)
        self.n_buckets = n_buckets
        self.always_reduce = always_reduce      # issue the collectives even at world_size 1 (single-GPU RCCL smoke test)
        self.scale_in_optimizer = scale_in_optimizer
        self._ranges = {}
        self._launch_streams = {}
        # measure=True: device events around the point where the backward's stream waits for the collectives - the part of the
        # all-reduces that did NOT hide under the backward (bench.py --gpus N reports it per rank); read with exposed_ms()
        self.measure = False
        self._exposed = []

    def world_size(self):
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def ranges_for(self, params, n_conv_layers):
        key = (n_conv_layers, tuple(p.numel() for p in params))
        if key not in self._ranges:
            self._ranges[key] = bucket_ranges([p.numel() for p in params], n_conv_layers, self.n_buckets)
        return self._ranges[key]

    def _launch_stream(self, device):
        if device not in self._launch_streams:
            self._launch_streams[device] = torch.cuda.Stream(device=device)
        return self._launch_streams[device]

    def run(self, engine, owner, noisy, params, out, grad_out, ws, grads, flat):
        nl = 2 * owner.n_layers + 1
        world = self.world_size()
        reduce = world > 1 or (self.always_reduce and (dist.is_initialized() or self.comm is not None))
        on_gpu = noisy.is_cuda
        pending = []
        if self.comm is not None and reduce:
            return self._run_native(engine, owner, noisy, params, out, grad_out, ws, grads, flat, world)
        for lb, le, fb, fe in self.ranges_for(params, nl):
            engine.backward(owner.n_layers, owner.channels_interval, noisy, params, out, grad_out, ws, grads,
                            layer_range=(lb, le), join=not (reduce and on_gpu))
            if not reduce:
                continue
            seg = flat[fb:fe]
            if on_gpu:
                main = torch.cuda.current_stream(noisy.device)
                launch = self._launch_stream(noisy.device)
                launch.wait_stream(main)                                    # the range's data-path kernels (BN, bias, head grads)
                with torch.cuda.stream(launch):
                    engine.join_weight_gradients(owner.n_layers, owner.channels_interval, noisy)   # ... and its weight gradients
                    work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            pending.append((work, seg))
        ev = self._mark(noisy) if (pending and on_gpu) else None
        for work, seg in pending:
            work.wait()                  # stream-level dependency on the RCCL stream, host does not block
            if not self.scale_in_optimizer:
                seg.mul_(1.0 / world)
        self._mark(noisy, ev)

    def _run_native(self, engine, owner, noisy, params, out, grad_out, ws, grads, flat, world):
        """The same schedule with the library's own RCCL entry: every bucket's all-reduce is enqueued on the launch stream C behind
        the events of S (data path) and W (weight gradients); S joins C once at the end.  Nothing here touches a process group
        or a host-side work handle, so the whole backward - collectives included - is capturable in the step's hipGraph."""
        nl = 2 * owner.n_layers + 1
        on_gpu = noisy.is_cuda
        main = torch.cuda.current_stream(noisy.device) if on_gpu else None
        launch = self._launch_stream(noisy.device) if on_gpu else None
        for lb, le, fb, fe in self.ranges_for(params, nl):
            engine.backward(owner.n_layers, owner.channels_interval, noisy, params, out, grad_out, ws, grads,
                            layer_range=(lb, le), join=not on_gpu)
            seg = flat[fb:fe]
            if on_gpu:
                launch.wait_stream(main)
                with torch.cuda.stream(launch):
                    engine.join_weight_gradients(owner.n_layers, owner.channels_interval, noisy)
                    self.comm.all_reduce_(seg)
                    if not self.scale_in_optimizer:
                        seg.mul_(1.0 / world)
            else:
                self.comm.all_reduce_(seg)
                if not self.scale_in_optimizer:
                    seg.mul_(1.0 / world)
        if on_gpu:
            ev = self._mark(noisy)
            main.wait_stream(launch)
            self._mark(noisy, ev)

    def _mark(self, noisy, first=None):
        if not self.measure or not noisy.is_cuda:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(noisy.device))
        if first is not None:
            self._exposed.append((first, e))
        return e

    def exposed_ms(self):
        """Per measured backward: the time torch's stream spent between reaching the join and getting past it (device events; call after
        a synchronize).  Clears the list."""
        out = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return out

    def describe(self, params, n_conv_layers):
        """What a scaling record should say about the exchange: world size as the transport reports it, the transport, the buckets in
        backward order (bytes)."""
        rs = self.ranges_for(params, n_conv_layers)
        if self.comm is not None:
            world = self.comm.engine.lib.wunet_comm_world(self.comm.handle)
            transport = "libwunet_hip wunet_comm_* (RCCL bound at run time)"
        else:
            world = dist.get_world_size(self.group) if dist.is_initialized() else 1
            transport = "torch.distributed " + (dist.get_backend(self.group) if dist.is_initialized() else "-")
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                    # noqa: BLE001 - a CPU build has none
            ver = None
        return {"world": world, "transport": transport, "rccl_version": ver, "bucket_bytes": [4 * (fe - fb) for _, _, fb, fe in rs],
                "bucket_layers": [[lb, le] for lb, le, _, _ in rs], "scale": "1/world in the Adam step" if self.scale_in_optimizer else "per bucket"}
