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


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "lang-seg_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import warnings
    warnings.simplefilter("ignore")
    import torch.distributed as dist
    from lseg_hip import dist as D
    from modules.models.lseg_net import LSegNet
    D.init_from_env("gloo")
    net = LSegNet(labels=["a", "b"], backbone="tiny16", features=64, arch_option=0, block_depth=0, activation="lrelu")
    params = [(n, p) for n, p in net.named_parameters() if p.dtype == torch.float32]
    g = torch.Generator().manual_seed(100 + rank)
    for i, (n, p) in enumerate(params):
        p.requires_grad_(True)
        p.grad = None if i % 17 == 3 else torch.randn(p.shape, generator=g)     # a few "unused" parameters
    mine = [None if p.grad is None else p.grad.clone() for _, p in params]
    cfg = net.cfg
    b = D.GradBucketer(params, cfg.depth, cfg.hooks)
    try:                                        # a bucket that was never readied is an error, not stale gradients
        b.finish()
        unreadied_raises = False
    except RuntimeError:
        unreadied_raises = True
    for i in range(len(b)):                     # same launch order on every rank (collectives are matched by order)
        b.ready(i)
    b.finish()
    out = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in params}
    keys = [ks for ks in b.keys]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    # second step after optimizer.zero_grad() with its default set_to_none=True (Lightning's default too): the bucket views are
    # gone, autograd allocates fresh .grad tensors -- the exchange must still reduce THIS step's values, not the stale flat buffers
    g2 = torch.Generator().manual_seed(200 + rank)
    for i, (n, p) in enumerate(params):
        p.grad = None
    for i, (n, p) in enumerate(params):
        if i % 13 != 5:
            p.grad = torch.randn(p.shape, generator=g2)
    mine2 = [None if p.grad is None else p.grad.clone() for _, p in params]
    for i in range(len(b)):
        b.ready(i)
    b.finish()
    out2 = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in params}
    gathered2 = [None] * world
    dist.all_gather_object(gathered2, mine2)
    if rank == 0:                               # verify here: only small python objects cross the process boundary
        bad = []
        for res, gat, tag in ((out, gathered, ""), (out2, gathered2, "#2")):
            for i, (n, _) in enumerate(params):
                if n.startswith("clip_pretrained."):      # frozen tower: in no optimizer group, in no bucket
                    continue
                g0 = gat[0][i] if gat[0][i] is not None else torch.zeros_like(res[n])
                g1 = gat[1][i] if gat[1][i] is not None else torch.zeros_like(res[n])
                if not torch.allclose(res[n], (g0 + g1) / 2, atol=1e-6):
                    bad.append(n + tag)
        q.put((bad, keys, len(params), unreadied_raises))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_gradient_allreduce_two_ranks():
    """The training exchange step (config 4): per-block buckets in backward order, asynchronous all-reduce, mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    bad, keys, nparams, unreadied_raises = q.get(timeout=150)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert not bad and nparams > 50, bad[:5]
    assert unreadied_raises
    # buckets in backward-completion order: DPT head + reassemble first, then ViT blocks depth-1 .. 0 (tiny16: depth 4),
    # each with the ProjectReadout hooked on it, the last one with the embeddings
    assert len(keys) == 5
    assert any("refinenet1" in k for k in keys[0]) and any("act_postprocess1.3" in k for k in keys[0])
    for j in range(4):
        assert all(f".blocks.{3 - j}." in k or "0.project.0" in k or j == 3 for k in keys[1 + j]), keys[1 + j][:4]
        assert any(f".blocks.{3 - j}." in k for k in keys[1 + j])
    assert any("patch_embed" in k for k in keys[4]) and any("pos_embed" in k for k in keys[4])


@pytest.mark.timeout(240)
def test_bench_gpus_2_without_torchrun_env_spawns_two_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must launch two ranks by itself (it used to time ONE
    device and print n_gpus 1); --launch-check runs the rendezvous and the line's reductions on gloo without touching a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["max_over_ranks"] == 2.0 and line["per_rank"] == [10.0, 20.0]


@pytest.mark.timeout(120)
def test_bench_refuses_to_measure_fewer_devices_than_asked():
    """No silent one-GPU run labelled as N: with fewer than N visible devices the launcher exits non-zero with the reason; a mismatching
    WORLD_SIZE is refused too."""
    import subprocess
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode != 0 and "refusing to measure fewer devices" in (r.stderr + r.stdout)
    env["WORLD_SIZE"] = "3"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode != 0
