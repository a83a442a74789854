"""N > 1 path on CPU: world_size-2 `gloo` processes.  Environment sharding needs no collective; the PPO-side
advantage normalisation (all-gather named by the north star, and the 3-scalar all-reduce) must equal the
single-process result on the concatenated shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rl-environment-for-component-placement_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcbenv import InstanceStream, env_seed, named_config
    from pcbenv.distributed import global_mean_std, normalize_advantages, shard_range
    g = torch.Generator().manual_seed(1234)
    full = torch.randn(world * 96, generator=g, dtype=torch.float32) * 3 + 0.7
    lo, hi = rank * 96, (rank + 1) * 96
    shard = full[lo:hi].clone()
    res = {}
    for mode in ("all_gather", "all_reduce"):
        n = normalize_advantages(shard, mode)
        ref = ((full - full.mean()) / full.std(unbiased=False))[lo:hi]
        res[mode] = float((n - ref).abs().max())
    first, last = shard_range(4)
    cfg = named_config("c2")
    res["h"] = [int(InstanceStream(cfg, env_seed(0, i)).next().comp_h[0]) for i in range(first, last)]
    res["range"] = (first, last)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_advantage_normalisation_and_sharding():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0]["all_gather"] < 1e-6 and out[1]["all_gather"] < 1e-6
    assert out[0]["all_reduce"] < 1e-5 and out[1]["all_reduce"] < 1e-5
    assert out[0]["range"] == (0, 4) and out[1]["range"] == (4, 8)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rl-environment-for-component-placement_amd"))
    from pcbenv import InstanceStream, env_seed, named_config
    cfg = named_config("c2")
    want = [int(InstanceStream(cfg, env_seed(0, i)).next().comp_h[0]) for i in range(8)]
    assert out[0]["h"] + out[1]["h"] == want  # the union of the shards is the single-process batch


def _module_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcbenv.distributed import allreduce_mean_, broadcast_module
    torch.manual_seed(100 + rank)  # every rank builds DIFFERENT weights and BatchNorm statistics
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 3), torch.nn.BatchNorm2d(3), torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(3 * 4 * 4, 5))
    net.train()
    net(torch.randn(7, 2, 6, 6))  # moves the running statistics
    before = torch.cat([t.detach().reshape(-1).float() for t in list(net.parameters()) + list(net.buffers())]).clone()
    broadcast_module(net)
    after = torch.cat([t.detach().reshape(-1).float() for t in list(net.parameters()) + list(net.buffers())])
    grads = [torch.full((3,), float(rank + 1)), torch.full((2, 2), float(10 * (rank + 1)))]
    allreduce_mean_(grads)
    out[rank] = {"before": before.numpy(), "after": after.numpy(), "g0": grads[0].numpy(), "g1": grads[1].numpy()}
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_model_broadcast_and_gradient_mean():
    """Data-parallel PPO starts every rank from rank 0's parameters AND buffers (BatchNorm running statistics) and
    averages gradients with one flat all-reduce (pcbenv/ppo.py)."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_module_worker, args=(world, port, out), nprocs=world, join=True)
    assert not np.array_equal(out[0]["before"], out[1]["before"])
    assert np.array_equal(out[0]["after"], out[0]["before"]) and np.array_equal(out[1]["after"], out[0]["before"])
    assert np.allclose(out[0]["g0"], 1.5) and np.allclose(out[1]["g0"], 1.5) and np.allclose(out[0]["g1"], 15.0)


def test_single_process_normalisation():
    from pcbenv.distributed import normalize_advantages
    a = torch.arange(10, dtype=torch.float32)
    n = normalize_advantages(a)
    assert abs(float(n.mean())) < 1e-6 and abs(float(n.std(unbiased=False)) - 1) < 1e-6
