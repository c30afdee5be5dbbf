"""N>1 path on CPU: two gloo ranks shard a global batch, run an (engine-stand-in) per-image
function on their slice only, and rank 0 gathers -- must equal the single-process result, with
no collective in the per-image work.  Also checks the bench timing reduction (max over ranks)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _per_image(x):          # stand-in for engine.forward: depends on the image only
    return torch.stack([x.mean(dim=(1, 2, 3)), x.amax(dim=(1, 2, 3)), (x * x).sum(dim=(1, 2, 3))], dim=1)


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from lseg_hip import dist as D
    r, lr, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((7, 3, 8, 8), generator=g)              # ragged: 7 images over 2 ranks
    mine = D.shard_batch(x, rank, world)
    out = _per_image(mine)
    sizes = [D.shard_range(7, i, world)[1] - D.shard_range(7, i, world)[0] for i in range(world)]
    full = D.gather_to_rank0(out, sizes)
    slow = D.max_over_ranks(1.0 + rank)
    cnt = D.sum_over_ranks(torch.tensor([float(mine.shape[0])]))
    D.barrier()
    if rank == 0:
        q.put((full, slow, cnt, _per_image(x)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharded_inference_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, slow, cnt, ref = q.get(timeout=90)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert torch.equal(full, ref)
    assert slow == 2.0 and cnt.item() == 7.0
