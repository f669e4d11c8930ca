"""Data-parallel gradient reduction for the PaSST step: one process per GPU, RCCL over xGMI.

The reference gets this from Lightning's DDP plugin (``trainer.accelerator=ddp``,
ex_audioset.py:488-489): NCCL ring all-reduce of 25 MB buckets driven by autograd hooks.  Here the
backward is an explicit kernel sequence (passt_amd.passt.passt_backward), so the reducer is driven
by its ``on_block_done`` callback instead: gradients live in ONE flat f32 buffer laid out in
``named_parameters()`` order; the buffer is cut into per-block buckets (head+final norm, block
depth-1 .. 0, patch-embed+positional+tokens) and each bucket's all-reduce (sum; the 1/world factor is
folded into the loss gradient) is launched the moment its last kernel is enqueued, overlapping the
rest of the backward.  xGMI is point-to-point (7 links x ~153 GB/s), so buckets are sized by block
(28 MB f32 for passt_s) -- large enough to run the links at bandwidth, small enough to overlap.

``torch.distributed`` is used as the transport (backend "nccl" == RCCL on ROCm; "gloo" on CPU for
the world_size-2 logic tests).  Parameters that never receive a gradient (``head_dist.*``,
SURVEY.md 2.4) are not part of the flat buffer at all, so there is no unused-parameter problem.
"""
import torch
import torch.distributed as dist

from . import ops


def bucket_layout(named_sizes, depth):
    """named_sizes: [(name, numel)] in flat-buffer order.  Returns {bucket_id: (start, end)} with
    bucket ids matching passt_backward's on_block_done(i): depth = head/final norm, 0..depth-1 =
    blocks, -1 = patch embedding / positional parameters / prefix tokens."""
    spans = {}
    off = 0
    for name, n in named_sizes:
        if name.startswith("blocks."):
            b = int(name.split(".")[1])
        elif name.startswith(("norm.", "head.")):
            b = depth
        else:
            b = -1
        s, e = spans.get(b, (off, off))
        if e != off and b in spans:
            raise ValueError(f"bucket {b} is not contiguous in the flat gradient buffer at {name}")
        spans[b] = (s if b in spans else off, off + n)
        off += n
    return spans


class RcclAbiTransport:
    """All-reduce through the library's own C-ABI entry points (pa_comm_init / pa_allreduce_bucket: RCCL resolved by the
    library, no torch.distributed in the data path).  The 128-byte communicator id is handed out through whatever
    torch.distributed group already exists (any backend; only ``broadcast_object_list`` is used, once).  Collectives
    run on a dedicated HIP stream, ordered after the producer kernels by an event and joined by ``wait()``."""

    def __init__(self, device, process_group=None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        lib = _lib.load()
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        box = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
            _lib.check(lib.pa_comm_unique_id(buf), "pa_comm_unique_id")
            box[0] = buf.raw
        if self.world > 1:
            # src is a GLOBAL rank: rank 0 of the (sub)group
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            dist.broadcast_object_list(box, src=src, group=process_group)
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            comm = C.c_void_p()
            _lib.check(lib.pa_comm_init(box[0], self.rank, self.world, C.byref(comm)), "pa_comm_init")
        self.comm = comm
        self.stream = torch.cuda.Stream(device=self.device, priority=-1)     # high priority: buckets start at the next kernel boundary

    def all_reduce(self, t):
        """In-place sum of the contiguous f32 / bf16 tensor ``t`` over the ranks; returns a handle with wait()."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        t.record_stream(self.stream)
        dt = self._lib.PA_F32 if t.dtype == torch.float32 else self._lib.PA_BF16
        self._lib.check(self._lib.load().pa_allreduce_bucket(self.comm, t.data_ptr(), t.numel(), dt, self.stream.cuda_stream),
                        "pa_allreduce_bucket")
        ev = torch.cuda.Event()
        ev.record(self.stream)
        dev = self.device

        class _Handle:
            def wait(self_inner):
                torch.cuda.current_stream(dev).wait_event(ev)
        return _Handle()

    def close(self):
        if getattr(self, "comm", None) is not None:
            self.stream.synchronize()
            self._lib.check(self._lib.load().pa_comm_destroy(self.comm), "pa_comm_destroy")
            self.comm = None

    def __del__(self):
        # never tear a communicator down from the garbage collector (peers may already have exited: ncclCommDestroy would
        # hang); owners call close() -- TrainStep.close(), ddp.detach().  A communicator still open here is a leak: say so.
        if getattr(self, "comm", None) is not None:
            import warnings
            warnings.warn("passt_amd.ddp: an RCCL communicator (C-ABI transport) was garbage-collected without close() -- call "
                          "TrainStep.close() / ddp.detach(net); the communicator and its stream stay allocated", ResourceWarning)


class GradReducer:
    """comm_dtype "fp32": all-reduce the f32 gradient bucket in place (exact sum).  "bf16": gradient compression for
    the wire -- the bucket is cast to bf16, all-reduced (RCCL sums in bf16) and added back as f32: half the bytes per
    link (172 MB instead of 345 MB per step for passt_s), at bf16 rounding of the exchanged sums (not the reference's
    behaviour: opt-in)."""

    def __init__(self, flat_grads, named_sizes, depth, process_group=None, comm_dtype="fp32", transport="torch", device=None):
        """transport "torch": torch.distributed collectives (backend nccl == RCCL on ROCm, gloo on CPU / in the tests);
        "rccl_abi": the library's C-ABI collective entry (RcclAbiTransport).  ``flat_grads`` may be None when the owner
        rebinds ``self.flat`` before every backward (the autograd path hands autograd a fresh buffer per step: attach());
        pass ``device`` then."""
        assert comm_dtype in ("fp32", "bf16") and transport in ("torch", "rccl_abi")
        self.flat = flat_grads
        self.device = torch.device(device) if device is not None else flat_grads.device
        self.total = sum(n for _, n in named_sizes)
        self.spans = bucket_layout(named_sizes, depth)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_dtype = comm_dtype
        self.transport = transport
        self.abi = RcclAbiTransport(self.device, process_group) if (transport == "rccl_abi" and self.world > 1) else None
        self.pending = []
        self._wires = {}
        self.timing = None

    def _all_reduce(self, t):
        if self.abi is not None:
            return self.abi.all_reduce(t)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def broadcast_(self, flat_params, src=0):
        """Make every rank's parameters equal to those of rank ``src`` OF THE GROUP (in place)."""
        if self.world > 1:
            gsrc = dist.get_global_rank(self.group, src) if self.group is not None else src
            dist.broadcast(flat_params, gsrc, group=self.group)

    def on_block_done(self, i):
        if self.world == 1:
            return
        s, e = self.spans[i]
        if e <= s:
            return
        # torch.distributed orders the collective after everything already enqueued on the current stream (the
        # kernels that produced this bucket) and runs it on the backend's communication stream
        ev0 = None
        if self.timing is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        if self.comm_dtype == "fp32":
            item = [self._all_reduce(self.flat[s:e]), None, s, e, i, ev0]
        else:
            # one conversion kernel of the library each way (pa_convert_f32 / pa_convert_to_f32), not torch's cast + copy_
            wire = self._wire(e - s)
            if self.flat.is_cuda:
                ops.convert_f32(self.flat[s:e], wire)
            else:                       # the gloo logic tests of this class run on CPU tensors
                wire.copy_(self.flat[s:e])
            item = [self._all_reduce(wire), wire, s, e, i, ev0]
        self.pending.append(item)

    def _wire(self, n):
        # wire buffers are reused from step to step (the reducer owns them: no allocator traffic inside the backward)
        key = (len(self.pending), n)
        buf = self._wires.get(key)
        if buf is None:
            buf = self._wires[key] = torch.empty(n, device=self.device, dtype=torch.bfloat16)
        return buf

    def _finish(self, item):
        w, wire, s, e, i, ev0 = item
        ev1 = None
        if self.timing is not None:
            # time the current stream spends blocked on this bucket = what the step actually pays for it
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
        w.wait()                      # current stream waits for the communication stream
        if self.timing is not None:
            ev2 = torch.cuda.Event(enable_timing=True)
            ev2.record()
            self.timing.append((i, (e - s) * (4 if wire is None else 2), ev0, ev1, ev2))
        if wire is not None:
            if self.flat.is_cuda:
                ops.convert_to_f32(wire, self.flat[s:e])
            else:
                self.flat[s:e].copy_(wire)

    def wait(self):
        pending, self.pending = self.pending, []
        for item in pending:
            self._finish(item)

    def drain(self):
        """Yield (start, end) of every launched bucket as soon as its all-reduce is ordered before the current stream.  If
        the consumer stops early (an exception in the per-bucket optimizer), the remaining collectives are still waited
        for: a half-drained reducer would leave all-reduces in flight on buffers the next step rewrites."""
        pending, self.pending = self.pending, []
        k = 0
        try:
            for k, item in enumerate(pending):
                self._finish(item)
                yield item[2], item[3]
            k = len(pending)
        finally:
            for item in pending[k + 1:] if k < len(pending) else ():
                self._finish(item)

    def start_timing(self):
        """bench.py, world > 1: record HIP events around every bucket (launch, first wait, released)."""
        self.timing = []

    def timing_summary(self):
        """[{bucket, bytes, in_flight_ms (launch -> released: includes overlap with the backward), exposed_wait_ms (current
        stream blocked), bus_GBps = 2(n-1)/n * bytes / in_flight}] of the buckets timed since start_timing(); call after a
        device synchronize."""
        out = []
        n = self.world
        for i, nbytes, ev0, ev1, ev2 in self.timing or []:
            inflight, exposed = ev0.elapsed_time(ev2), ev1.elapsed_time(ev2)
            out.append({"bucket": i, "bytes": nbytes, "in_flight_ms": round(inflight, 4), "exposed_wait_ms": round(exposed, 4),
                        "bus_GBps": round(2.0 * (n - 1) / n * nbytes / max(inflight, 1e-6) / 1e6, 1)})
        return out

    def close(self):
        if self.abi is not None:
            self.abi.close()
            self.abi = None

    def launch_order(self):
        """bucket ids in the order the backward completes them: head, blocks depth-1 .. 0, patch embedding"""
        return sorted(self.spans, key=lambda b: (b != max(self.spans), -b))

    def comm_info(self):
        """What the communicator itself reports, for bench.py's N > 1 line: {backend, rccl_version, nranks, rank}.  C-ABI
        transport: ncclGetVersion / ncclCommCount / ncclCommUserRank of the live communicator (pa_comm_info);
        torch transport: the process group's backend and size, RCCL version from torch.cuda.nccl."""
        info = {"transport": self.transport, "nranks": self.world, "rank": dist.get_rank(self.group) if dist.is_initialized() else 0}
        if self.abi is not None:
            import ctypes as C
            from . import _lib
            v, n, r, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            _lib.check(_lib.load().pa_comm_info(self.abi.comm, C.byref(v), C.byref(n), C.byref(r), C.byref(d)), "pa_comm_info")
            info.update(backend="rccl (C ABI: pa_comm_*)", rccl_version=v.value, nranks=n.value, rank=r.value, device=d.value,
                        source="ncclGetVersion / ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the live communicator")
        elif dist.is_initialized():
            be = dist.get_backend(self.group)
            info.update(backend=be, source="torch.distributed process group")
            if be == "nccl":
                try:
                    ver = torch.cuda.nccl.version()
                    info["rccl_version"] = ver[0] * 10000 + ver[1] * 100 + ver[2] if isinstance(ver, tuple) else int(ver)
                except Exception as e:        # noqa: BLE001 -- information only
                    info["rccl_version"] = f"unavailable: {e}"
        return info

    def measure_idle(self, repeats=3):
        """Pure all-reduce of the same buckets with the compute stream idle (nothing else on the GPU): what the wire alone
        costs, next to timing_summary()'s in-step figures (which include queueing behind the backward).  Works on a scratch
        buffer; [{bucket, bytes, ms (best of `repeats`), bus_GBps}] in launch order.  Call on every rank."""
        if self.world == 1:
            return []
        es = 4 if self.comm_dtype == "fp32" else 2
        buf = torch.zeros(max(e - s for s, e in self.spans.values()), device=self.device,
                          dtype=torch.float32 if es == 4 else torch.bfloat16)
        out = []
        for b in self.launch_order():
            s, e = self.spans[b]
            if e <= s:
                continue
            best = None
            for _ in range(repeats + 1):            # the first pass warms the channel up
                if dist.is_initialized():
                    dist.barrier(group=self.group)
                torch.cuda.synchronize(self.device)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                self._all_reduce(buf[:e - s]).wait()
                ev1.record()
                torch.cuda.synchronize(self.device)
                ms = ev0.elapsed_time(ev1)
                best = ms if best is None else min(best, ms)
            nbytes = (e - s) * es
            out.append({"bucket": b, "bytes": nbytes, "ms": round(best, 4),
                        "bus_GBps": round(2.0 * (self.world - 1) / self.world * nbytes / max(best, 1e-6) / 1e6, 1)})
        return out

    def __del__(self):
        # communicators are NOT destroyed from the garbage collector: at interpreter shutdown the peers may be gone
        # already (ncclCommDestroy then hangs, and a hang is not an exception).  close() is explicit.
        pass

    def bucket_bytes(self):
        """{bucket id: bytes on the wire} in launch order (head, blocks depth-1 .. 0, patch embedding)."""
        es = 4 if self.comm_dtype == "fp32" else 2
        return {b: (self.spans[b][1] - self.spans[b][0]) * es for b in self.launch_order()}


# ------------------------------------------------------------------------------------------------------------------
# The drop-in (autograd) path: net(x); loss.backward() as an unmodified ex_audioset.py runs it
# ------------------------------------------------------------------------------------------------------------------
def grad_layout(net):
    """[(name, numel)] of the parameters the backward writes, in named_parameters() order (head_dist.* never receives a
    gradient: models/passt.py:583-595 does not use it) -- the layout of the flat gradient buffer of both paths."""
    return [(n, p.numel()) for n, p in net.named_parameters() if not n.startswith("head_dist.")]


def attach(net, process_group=None, comm_dtype="fp32", transport="torch", broadcast=True):
    """Data parallelism for the AUTOGRAD path without a DistributedDataParallel wrapper -- the three-line change to
    ex_audioset.py (reference: Lightning's DDP plugin, ex_audioset.py:488-489).

    PaSST is ONE autograd node here, so torch's DDP hooks would see every parameter gradient at the same instant, after the
    last kernel of the backward: the whole 345 MB all-reduce exposed.  After ``attach`` the node's backward does the
    reduction itself: gradients are produced into one flat buffer, each block's bucket is all-reduced (sum of gradients
    pre-scaled by 1/world = DDP's mean) the moment its last kernel is enqueued -- overlapping the rest of the backward --
    and the backward returns only after the current stream is ordered behind the last bucket.  ``loss.backward()`` then
    leaves averaged ``.grad``s exactly as DDP would; any torch optimizer follows.  Like DDP's constructor, ``attach``
    broadcasts rank 0's parameters.  Do not ALSO wrap the module in DistributedDataParallel.  Returns the reducer
    (``.close()`` / ``detach(net)`` releases a C-ABI communicator)."""
    dev = next(net.parameters()).device
    red = GradReducer(None, grad_layout(net), len(net.blocks), process_group, comm_dtype=comm_dtype, transport=transport, device=dev)
    if broadcast and red.world > 1:
        src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
        with torch.no_grad():
            for p in net.parameters():
                dist.broadcast(p.data, src, group=process_group)
        net.mark_params_updated()
    object.__setattr__(net, "_ddp", red)
    if red.world > 1:
        # the all-reduce kernels share the CUs with the backward: GEMMs go out one work item per workgroup
        import os
        if os.environ.get("PASST_AMD_DDP_PERSISTENT") != "1":
            from . import _lib
            # OR the bit in (a TrainStep or an A/B knob may have set others) and remember that WE set it, for detach()
            red._set_no_persist = not (getattr(net, "_gemm_flags", 0) & _lib.GEMM_NO_PERSIST)
            object.__setattr__(net, "_gemm_flags", getattr(net, "_gemm_flags", 0) | _lib.GEMM_NO_PERSIST)
    return red


def detach(net):
    red = getattr(net, "_ddp", None)
    if red is not None:
        red.wait()
        red.close()
    object.__setattr__(net, "_ddp", None)
    if red is not None and getattr(red, "_set_no_persist", False):
        from . import _lib
        object.__setattr__(net, "_gemm_flags", getattr(net, "_gemm_flags", 0) & ~_lib.GEMM_NO_PERSIST)
        red._set_no_persist = False
