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


class GradReducer:
    def __init__(self, flat_grads, named_sizes, depth, process_group=None):
        self.flat = flat_grads
        self.spans = bucket_layout(named_sizes, depth)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.pending = []

    def on_block_done(self, i):
        if self.world == 1:
            return
        s, e = self.spans[i]
        if e > s:
            # torch.distributed orders the collective after everything already enqueued on the
            # current stream (the kernels that produced this bucket) and runs it on RCCL's stream
            self.pending.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()                      # current stream waits for the communication stream
        self.pending = []
