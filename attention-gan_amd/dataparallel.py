"""Data-parallel gradient exchange: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

New component (the reference is single-GPU).  Every replica keeps BatchNorm statistics and the
DAMSM B x B matrix local, exactly as the reference computes them per batch; the only exchange is a SUM all-reduce of
weight gradients, issued per bucket on a side stream as soon as backward has produced every gradient of the bucket,
and joined right before that optimiser's step (which applies the 1/world_size scale inside the fused Adam kernel).

Buckets are contiguous slices of FlatAdam's flat gradient buffer, so a bucket is one large RCCL call; xGMI is
point-to-point (7 links x ~153 GB/s per GPU), so few large messages beat many small ones.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .backend import functional as HF
from .optim import FlatAdam


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank_of(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def any_rank(flag: bool, group=None, device=None) -> bool:
    """True on EVERY rank if `flag` is true on ANY rank: one MAX all-reduce of a single int32.  The data-parallel form of a
    per-batch decision that changes the collectives a rank issues -- the reference's batch guard `if min(lengths) < 2 or
    len(words) < BATCH_SIZE: continue` (train.py:112): a rank that skipped a step on its own would leave the others waiting in
    the gradient all-reduce.  With one rank it is `bool(flag)`, no communication."""
    if world_size(group) == 1:
        return bool(flag)
    dev = device if (device is not None and dist.get_backend(group) == "nccl") else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


@torch.no_grad()
def sync_buffers_(module: torch.nn.Module, group=None) -> None:
    """Average the floating-point buffers (BatchNorm running statistics) over the replicas, in place; integer buffers
    (num_batches_tracked) are identical on every rank by construction.  Called before a checkpoint is written."""
    w = world_size(group)
    if w == 1:
        return
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(w)
    o = 0
    for b in bufs:
        b.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()


@torch.no_grad()
def averaged_buffers(module: torch.nn.Module, group=None) -> dict:
    """{buffer name: mean over the replicas} for the floating-point buffers of `module`, WITHOUT touching the live buffers (a
    checkpoint must not change the run it is taken from).  A collective: every rank of `group` must call it."""
    w = world_size(group)
    named = [(k, b) for k, b in module.named_buffers() if b.is_floating_point()]
    if w == 1 or not named:
        return {k: b.detach().clone() for k, b in named}
    flat = torch.cat([b.reshape(-1) for _, b in named])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(w)
    out, o = {}, 0
    for k, b in named:
        out[k] = flat[o:o + b.numel()].view_as(b).clone()
        o += b.numel()
    return out


def broadcast_module_(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Identical initial weights and BN buffers on every replica."""
    if world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class DirectRccl:
    """One RCCL communicator per process group, driven through the C ABI (include/agan.h: agan_comm_* / agan_allreduce_bucket): each
    bucket goes out as reduce-scatter + all-gather on the communicator's stream.  Opt-in (AGAN_RCCL_DIRECT=1); the default exchange
    is torch.distributed's all_reduce, which is also what the CPU (gloo) tests can run.  The id travels over the existing process
    group, whatever its backend.  PARITY UNPINNED across ranks: no multi-GPU box has run it yet (one-rank identity test and the
    host-side chunk arithmetic test only) -- DESIGN.md section 6."""

    _shared: dict = {}          # keyed by the group OBJECT (kept alive by the key: no id() reuse after garbage collection)

    def __init__(self, group=None):
        import ctypes
        from .backend import lib as L
        self.L = L
        self.group = group
        rank, world = rank_of(group), world_size(group)
        ident = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(L.COMM_ID_BYTES)
            L.call("agan_comm_unique_id", buf)
            ident[0] = bytes(buf.raw)
        if world > 1:
            # `src` of broadcast_object_list is a GLOBAL rank: group rank 0 of a subgroup need not be global rank 0
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(ident, src=src, group=group)
        self._id = ctypes.create_string_buffer(ident[0], L.COMM_ID_BYTES)
        self.comm = ctypes.c_void_p()
        L.call("agan_comm_init", ctypes.byref(self.comm), rank, world, self._id)
        # ONE stream for every collective of this communicator (the generator's and the three discriminators' buckets): RCCL orders
        # the operations of a communicator by issue order, and one stream makes that order explicit on the device as well
        self.stream = torch.cuda.Stream(priority=int(os.environ.get("AGAN_DP_COMM_PRIO", "0")))     # (normal priority: see GradBuckets)

    @classmethod
    def get(cls, group=None):
        if group not in cls._shared:
            if not cls._shared:
                import atexit
                atexit.register(cls.close_all)          # communicators are destroyed before the interpreter tears the library down
            cls._shared[group] = cls(group)
        return cls._shared[group]

    @classmethod
    def close_all(cls) -> None:
        for c in list(cls._shared.values()):
            try:
                c.close()
            except Exception:      # interpreter shutdown: the library may already be gone
                pass
        cls._shared.clear()

    def all_reduce_(self, chunk: torch.Tensor, stream: torch.cuda.Stream, bf16: bool = False) -> None:
        """SUM all-reduce of one fp32 bucket on `stream`; bf16=True: 16-bit wire format, fp32 accumulation on arrival
        (include/agan.h: agan_allreduce_bucket_dt with AGAN_DT_BF16 -- the same formula as all_reduce_bf16_)."""
        import ctypes
        if not bf16:
            self.L.call("agan_allreduce_bucket", self.comm, ctypes.c_void_p(chunk.data_ptr()), chunk.numel(), ctypes.c_void_p(stream.cuda_stream))
            return
        world = world_size(self.group)
        need = self.L.load().agan_allreduce_scratch_bytes(chunk.numel(), world, self.L.DT_BF16)
        with torch.cuda.stream(stream):              # scratch from the caching allocator, owned by the comm stream
            scratch = torch.empty(need, dtype=torch.uint8, device=chunk.device)
        self.L.call("agan_allreduce_bucket_dt", self.comm, ctypes.c_void_p(chunk.data_ptr()), chunk.numel(), self.L.DT_BF16,
                    ctypes.c_void_p(scratch.data_ptr()), need, ctypes.c_void_p(stream.cuda_stream))

    def close(self) -> None:
        if self.comm:
            torch.cuda.synchronize()
            self.L.call("agan_comm_destroy", self.comm)
            self.comm = None


def all_reduce_bf16_(chunk: torch.Tensor, group=None) -> None:
    """SUM all-reduce of an fp32 gradient bucket that moves 16-bit values and ACCUMULATES IN FP32 on arrival (AGAN_DP_BF16=1):
    half the bytes of the fp32 reduce-scatter + all-gather on the per-link-bound xGMI rings.  In place, on the current stream.

        1. every rank rounds its bucket to bf16                                   (agan_exchange_pack_bf16);
        2. all-to-all: rank j receives piece j of every rank and sums the W pieces in rank order in fp32, rounding once
           (agan_exchange_sum_bf16: no bf16 accumulation, the error does not grow with the number of ranks' additions);
        3. the sums go back as bf16 (all-gather) and are widened                  (agan_exchange_unpack_bf16):
                   result = fp32(bf16(sum_r fp32(bf16(g_r)))).

    Two roundings of 2^-9 relative each -- far below the gradient noise of a 24-image batch; Adam's sign-like first steps do not see
    it (tests: 2-rank step parity at 1e-3).  On device tensors the three element-wise passes are HIP kernels behind the C ABI
    (include/agan.h: agan_exchange_*; one launch each) and only the two collectives are torch.distributed's.  The plain-torch
    branch below exists for the CPU tests (gloo has no all-to-all on these tensors: an all-gather of the whole rounded bucket, the
    same sums in the same order -- identical values, more bytes)."""
    w = world_size(group)
    n = chunk.numel()
    if chunk.is_cuda and n % 4 == 0 and chunk.data_ptr() % 16 == 0 and (w == 1 or dist.get_backend(group) == "nccl"):
        import ctypes
        from .backend import lib as L
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        nw = L.load().agan_exchange_wire_elems(n, w)
        per = nw // w
        send = torch.empty(nw, dtype=torch.bfloat16, device=chunk.device)
        L.call("agan_exchange_pack_bf16", ctypes.c_void_p(chunk.data_ptr()), ctypes.c_void_p(send.data_ptr()), n, nw, st)
        if w > 1:
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=group)
        else:
            recv = send
        mine = torch.empty(per, dtype=torch.bfloat16, device=chunk.device)
        L.call("agan_exchange_sum_bf16", ctypes.c_void_p(recv.data_ptr()), w, per, ctypes.c_void_p(mine.data_ptr()), st)
        if w > 1:
            dist.all_gather_into_tensor(send, mine, group=group)          # the send image is free again: it receives the sums
        else:
            send = mine
        L.call("agan_exchange_unpack_bf16", ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(chunk.data_ptr()), n, st)
        return
    if w == 1:
        chunk.copy_(chunk.to(torch.bfloat16).float().to(torch.bfloat16).float())
        return
    per = (n + w - 1) // w
    send = torch.zeros(per * w, dtype=torch.bfloat16, device=chunk.device)
    send[:n].copy_(chunk)
    parts = [torch.empty_like(send) for _ in range(w)]
    dist.all_gather(parts, send, group=group)
    acc = parts[0].float()
    for r in range(1, w):                           # rank order, fp32
        acc += parts[r].float()
    chunk.copy_(acc.to(torch.bfloat16)[:n])


class GradBuckets:
    """Bucketed, backward-overlapped all-reduce over a FlatAdam's flat gradient buffer."""

    def __init__(self, opt: FlatAdam, bucket_bytes: int = 64 << 20, group=None):
        self.opt, self.group = opt, group
        self.world = world_size(group)
        self.bounds: List[tuple] = []          # (start, end) element ranges of the flat gradient buffer
        self.param_bucket: List[int] = []
        per = max(1, bucket_bytes // 4)
        start, b = 0, 0
        for p, o in zip(opt.params, opt.offsets):
            end = o + (p.numel() + 3) // 4 * 4
            if end - start > per and o > start:
                self.bounds.append((start, o))
                start, b = o, b + 1
            self.param_bucket.append(b)
        self.bounds.append((start, opt.numel))
        self.counts = [self.param_bucket.count(i) for i in range(len(self.bounds))]
        self.members = [[i for i, b in enumerate(self.param_bucket) if b == k] for k in range(len(self.bounds))]
        self._pending = list(self.counts)
        self._handles: List = []
        self._armed = False
        self.comm_stream: Optional[torch.cuda.Stream] = None
        # AGAN_DP_FORCE=1: run the whole exchange machinery (hooks, comm stream, async handles) in a world of ONE rank as well --
        # the rehearsal of the `nccl` backend that a one-GPU box allows (RCCL refuses two ranks on one device)
        self.active = self.world > 1 or (os.environ.get("AGAN_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized())
        # AGAN_DP_BF16=1: exchange 16-bit values with fp32 accumulation on arrival (all_reduce_bf16_) instead of the fp32 all-reduce
        self.bf16 = os.environ.get("AGAN_DP_BF16") == "1"
        # AGAN_RCCL_DIRECT=1: the library's own communicator (agan_allreduce_bucket[_dt]) instead of torch.distributed's collectives;
        # combined with AGAN_DP_BF16 it runs the 16-bit wire format on that communicator (grouped send/recv + all-gather)
        self.direct = DirectRccl.get(group) if (self.active and opt.flat.is_cuda and os.environ.get("AGAN_RCCL_DIRECT") == "1") else None
        # optional measurement (bench.py): HIP events around every wait of a compute stream on the exchange -> exposed_ms()
        self.measure = False
        self._waits: List = []
        if self.active:
            if self.direct is not None:
                self.comm_stream = self.direct.stream
            elif opt.flat.is_cuda:
                # NORMAL priority (AGAN_DP_COMM_PRIO overrides).  Round 2 made this a high-priority stream "ahead of the chip-filling
                # compute kernels"; measured in round 3 (profiles/r03_dp_rehearsal.txt): on this ROCm stack a priority -1 stream that
                # merely takes part in the event waits between the three discriminator streams costs the step 16 ms (42.9 vs 26.7 ms,
                # even with the collective itself stubbed out) -- cross-priority waits serialise the hardware queues.  At normal
                # priority the whole exchange machinery costs 1.0-1.4 ms per step in the one-rank rehearsal.
                self.comm_stream = torch.cuda.Stream(device=opt.flat.device, priority=int(os.environ.get("AGAN_DP_COMM_PRIO", "0")))
            # AGAN_DP_NO_HOOKS=1: no per-parameter hooks -- every bucket is exchanged in finish(), after the whole backward (no overlap
            # with this optimiser's own backward; the other optimisers' work still overlaps it)
            if os.environ.get("AGAN_DP_NO_HOOKS") != "1":
                for i, p in enumerate(opt.params):
                    p.register_post_accumulate_grad_hook(self._make_hook(i))

    def arm(self) -> None:
        """Call before the backward whose gradients should be exchanged (after zero_grad)."""
        self._pending = list(self.counts)
        self._handles = []
        self._armed = True

    def _make_hook(self, i: int):
        def hook(_param):
            if not self._armed:
                return
            b = self.param_bucket[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b: int) -> None:
        s, e = self.bounds[b]
        self.opt._rebind(self.members[b])      # only this bucket's parameters: the others may still be mid-backward
        for i in self.members[b]:              # from here to zero_grad() no backward kernel may write into this bucket (functional._grad_out)
            self.opt.params[i]._agan_grad_dst.locked = True
        chunk = self.opt.grad[s:e]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            HF.join_all_side_streams(self.comm_stream)          # weight gradients forked onto side streams
            if self.direct is not None:
                chunk.record_stream(self.comm_stream)
                self.direct.all_reduce_(chunk, self.comm_stream, self.bf16)      # joined in finish() through the comm stream
                return
            if os.environ.get("AGAN_DP_STUB") == "1":          # (measurement knob: the stream choreography without the collective)
                return
            with torch.cuda.stream(self.comm_stream):
                if self.bf16:
                    all_reduce_bf16_(chunk, self.group)       # several ops, all on the comm stream; joined through it in finish()
                else:
                    self._handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        elif self.bf16:
            all_reduce_bf16_(chunk, self.group)
        else:
            self._handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def exchange_all(self) -> float:
        """Exchange EVERY bucket now, after the backward has completed on the current stream (no hooks, no overlap with this
        optimiser's own backward): the form the segment-wise captured step uses between its graphs (GanTrainStep.capture_segments),
        where the backward is a replayed HIP graph and Python hooks never run.  The collectives go out on the comm stream in bucket
        order; the current stream waits for them.  Returns the scale the optimiser must apply (1/world_size)."""
        if not self.active:
            return 1.0
        cur = torch.cuda.current_stream() if self.opt.flat.is_cuda else None
        handles = []
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(cur)
        for s, e in self.bounds:
            chunk = self.opt.grad[s:e]
            if self.direct is not None:
                self.direct.all_reduce_(chunk, self.comm_stream, self.bf16)
            elif self.comm_stream is not None:
                with torch.cuda.stream(self.comm_stream):
                    if self.bf16:
                        all_reduce_bf16_(chunk, self.group)
                    else:
                        handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            elif self.bf16:
                all_reduce_bf16_(chunk, self.group)
            else:
                handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._join(cur, handles)
        return 1.0 / self.world

    @property
    def mode(self) -> str:
        """what moves the buckets: for the bench line / logs"""
        if not self.active:
            return "none (one rank)"
        wire = "bf16 wire, fp32 accumulate" if self.bf16 else "fp32"
        how = "agan_allreduce_bucket (library communicator: reduce-scatter + all-gather)" if self.direct is not None else \
              ("torch.distributed all_to_all + all_gather" if self.bf16 else "torch.distributed all_reduce")
        return f"{how}, {wire}"

    def _join(self, cur, handles) -> None:
        """the compute stream `cur` waits for the exchange (async handles: Work.wait() makes the CURRENT stream wait for the
        collective; then the comm stream itself); with `measure` on, two events bracket the waits -- their distance is the time the
        stream sat blocked = EXPOSED communication, nothing else being queued between them"""
        measure = self.measure and cur is not None
        if measure:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(cur)
        for h in handles:
            h.wait()
        if self.comm_stream is not None:
            cur.wait_stream(self.comm_stream)
        if measure:
            b.record(cur)
            self._waits.append((a, b))

    def exposed_ms(self) -> float:
        """sum of the measured waits since the last call (synchronises the device)"""
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._waits)
        self._waits = []
        return total

    def finish(self) -> float:
        """Join outstanding exchanges; returns the scale the optimiser must apply (1/world_size)."""
        if not self.active:
            return 1.0
        # buckets whose parameters received no gradient this backward (unused params) are reduced too so that
        # every rank issues the same collectives
        for b, left in enumerate(self._pending):
            if self._armed and left > 0:
                self._pending[b] = 0
                self._launch(b)
        self._join(torch.cuda.current_stream() if self.opt.flat.is_cuda else None, self._handles)
        self._handles, self._armed = [], False
        return 1.0 / self.world
