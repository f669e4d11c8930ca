"""world_size-2 test of the data-parallel gradient reducer on CPU (gloo): bucket layout follows the
backward's completion order, every bucket is all-reduced exactly once, the result equals the sum over
ranks, and parameters that never get a gradient (head_dist.*) are not in the buffer."""
import os
import socket
import warnings

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from passt_amd.ddp import GradReducer, bucket_layout


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_model():
    import passt_amd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return passt_amd.PaSST(img_size=(128, 100), stride=10, embed_dim=128, depth=3, num_heads=2, num_classes=7,
                               distilled=True)


def test_bucket_layout_partitions_the_flat_buffer():
    m = _small_model()
    names = m._grad_names
    sizes = [(n, p.numel()) for n, p in m.named_parameters() if n in names]
    assert not any(n.startswith("head_dist") for n, _ in sizes)
    spans = bucket_layout(sizes, 3)
    assert sorted(spans) == [-1, 0, 1, 2, 3]
    order = [-1, 0, 1, 2, 3]
    pos = 0
    for b in order:
        s, e = spans[b]
        assert s == pos and e > s
        pos = e
    assert pos == sum(n for _, n in sizes) == m._n_grad_elems


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _small_model()
    names = m._grad_names
    sizes = [(n, p.numel()) for n, p in m.named_parameters() if n in names]
    total = sum(n for _, n in sizes)
    flat = torch.arange(total, dtype=torch.float32) * (rank + 1)
    red = GradReducer(flat, sizes, 3)
    seen = []
    for blk in (3, 2, 1, 0, -1):          # the order passt_backward reports completion
        red.on_block_done(blk)
        seen.append(len(red.pending))
    red.wait()
    expect = torch.arange(total, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = bool(torch.equal(flat, expect))
    # drain(): buckets come back in launch order and cover the whole buffer exactly once (TrainStep updates per bucket)
    flat2 = torch.ones(total) * (rank + 1)
    red2 = GradReducer(flat2, sizes, 3)
    for blk in (3, 2, 1, 0, -1):
        red2.on_block_done(blk)
    spans = list(red2.drain())
    ok = ok and spans == [red2.spans[b] for b in (3, 2, 1, 0, -1)] and sorted(spans)[0][0] == 0 and \
        sum(e - s for s, e in spans) == total and bool(torch.equal(flat2, torch.full((total,), 3.0)))
    # bf16 wire format: sums of values exactly representable in bf16 stay exact
    flat3 = torch.full((total,), 0.5 * (rank + 1))
    red3 = GradReducer(flat3, sizes, 3, comm_dtype="bf16")
    for blk in (3, 2, 1, 0, -1):
        red3.on_block_done(blk)
    red3.wait()
    ok = ok and bool(torch.equal(flat3, torch.full((total,), 1.5))) and sum(red3.bucket_bytes().values()) == 2 * total
    # broadcast_: every rank ends with rank 0's parameters
    pbuf = torch.full((16,), float(rank + 7))
    red.broadcast_(pbuf)
    ok = ok and bool(torch.equal(pbuf, torch.full((16,), 7.0)))
    # attach(): the autograd path's reducer -- rank 0's parameters everywhere (DDP's constructor does the same), a reducer
    # whose flat buffer is rebound before every backward, per-model GEMM launch flags, nothing of it deep-copied (SWA)
    import copy
    from passt_amd import _lib, ddp
    torch.manual_seed(100 + rank)
    net = _small_model()                        # differently initialised per rank
    red4 = ddp.attach(net)
    first = next(net.parameters()).detach().clone()
    gathered = [torch.empty_like(first) for _ in range(world)]
    dist.all_gather(gathered, first)
    ok = ok and all(bool(torch.equal(g, gathered[0])) for g in gathered)
    ok = ok and net._ddp is red4 and net._gemm_flags == _lib.GEMM_NO_PERSIST and red4.total == total
    ok = ok and [n for n, _ in ddp.grad_layout(net)] == [n for n, _ in sizes]
    twin = copy.deepcopy(net)
    ok = ok and twin._ddp is None and twin._gemm_flags == 0
    for step in range(2):                       # what _PasstFunction.backward does: fresh buffer, buckets in order, wait
        red4.flat = torch.full((total,), float(rank + 1 + step))
        for blk in (3, 2, 1, 0, -1):
            red4.on_block_done(blk)
        red4.wait()
        ok = ok and bool(torch.equal(red4.flat, torch.full((total,), float(3 + 2 * step))))
    info = red4.comm_info()
    ok = ok and info["nranks"] == world and info["rank"] == rank and info["backend"] == "gloo"
    ddp.detach(net)
    ok = ok and net._ddp is None and net._gemm_flags == 0
    q.put((rank, ok, seen, red.world))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, seen, w in res:
        assert ok, f"rank {rank}: reduced gradients differ from the sum over ranks"
        assert seen == [1, 2, 3, 4, 5] and w == 2
