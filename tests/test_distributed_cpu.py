"""CPU (gloo, world_size 2) test of the N>1 path: utterance sharding + the all-gather of
embeddings (BASELINE config 5 / SURVEY §8e).  The device kernels are not involved: every rank
fabricates the embeddings its utterances would produce, so the test pins partitioning,
padding and re-ordering."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _emb_of(i, stream):
    g = torch.Generator().manual_seed(1000 * i + stream)
    return torch.randn(192, generator=g)


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from targetdiarization_amd.pipeline import gather_embeddings, shard_indices
    mine = shard_indices(n_total, rank, world)
    local = torch.stack([_emb_of(i, s) for i in mine for s in (0, 1)]) if mine else torch.zeros(0, 192)
    full = gather_embeddings(local, n_total, rank, world)
    ref = torch.stack([_emb_of(i, s) for i in range(n_total) for s in (0, 1)])
    q.put((rank, bool(torch.equal(full, ref)), tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8, 1])
def test_shard_and_allgather_world2(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (n_total * 2, 192), (rank, ok, shape)


def test_shard_indices_partition():
    from targetdiarization_amd.pipeline import shard_indices
    for n in (0, 1, 7, 1000):
        for world in (1, 2, 4, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _spk_emb(i, stream):
    """utterance i, stream s -> a noisy copy of one of three "speaker" centroids (what config 5's clustering consumes)"""
    g = torch.Generator().manual_seed(77 + (2 * i + stream) % 3)
    c = torch.randn(192, generator=g)
    g2 = torch.Generator().manual_seed(1000 * i + stream)
    return c + 0.05 * torch.randn(192, generator=g2)


def _cluster_worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from targetdiarization_amd.pipeline import cluster_embeddings, gather_embeddings, shard_indices
    mine = shard_indices(n_total, rank, world)
    local = torch.stack([_spk_emb(i, s) for i in mine for s in (0, 1)])
    labels = cluster_embeddings(gather_embeddings(local, n_total, rank, world))
    q.put((rank, [int(x) for x in labels]))
    dist.barrier()
    dist.destroy_process_group()


def test_clustering_of_the_gathered_embeddings_world2():
    """the all-gather feeds the clustering (BASELINE config 5): every rank gets the labels a single rank would compute"""
    from targetdiarization_amd.pipeline import cluster_embeddings
    world, n_total = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_cluster_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [int(x) for x in cluster_embeddings(torch.stack([_spk_emb(i, s) for i in range(n_total) for s in (0, 1)]))]
    assert res[0] == want and res[1] == want
    # three speakers -> three clusters, every embedding with its centroid's group
    groups = {}
    for j, lab in enumerate(want):
        groups.setdefault(j % 3, set()).add(lab)
    assert all(len(v) == 1 and -1 not in v for v in groups.values()) and len({next(iter(v)) for v in groups.values()}) == 3
