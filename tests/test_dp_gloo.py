"""world_size-2 (gloo, CPU) test of the data-parallel step's mathematics and control plane:
independent learners, each with its own minibatch, ONE all-reduce(sum) of the flat gradient, RMSProp with divisor
R*B  ==  a single learner fed the concatenated R*B minibatch.  On the GPU the all-reduce is RCCL inside
libsdqn_hip (sdqn_dp_init); here gloo stands in for it and the oracle for the kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.dqn_numpy import OracleDQN, xavier_weights
from util import random_minibatch

A, B, R = 4, 4, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=R)
    # control plane used by bench.py: rank 0 creates a 128-byte id, everyone receives it
    ids = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    assert ids[0] == bytes(range(128))
    net = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 5))       # identical replicas
    for step in range(3):
        mb = random_minibatch(B, A, 100 + 10 * step + rank)             # own experience per learner
        g, cost, _, _ = net.gradients(mb)
        flat = torch.from_numpy(np.concatenate([x.ravel() for x in g]))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)                     # the ONE collective of the step
        off, gs = 0, []
        for x in g:
            gs.append(flat[off:off + x.size].numpy().reshape(x.shape))
            off += x.size
        net.rmsprop(gs, R * B)
    # replicas stay bit-identical
    w = torch.from_numpy(np.concatenate([x.ravel() for x in net.W]))
    ws = [torch.empty_like(w) for _ in range(R)]
    dist.all_gather(ws, w)
    assert all(torch.equal(ws[0], x) for x in ws)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)                  # max-over-ranks timing reduce
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == float(R)
    if rank == 0:
        np.save(out_path, w.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_dp_mean_equals_concatenated_batch(tmp_path):
    out = str(tmp_path / "w.npy")
    mp.spawn(_worker, args=(_free_port(), out), nprocs=R, join=True)
    got = np.load(out)
    ref = OracleDQN(A, batch_size=R * B, weights=xavier_weights(A, 5))
    for step in range(3):
        mbs = [random_minibatch(B, A, 100 + 10 * step + r) for r in range(R)]
        cat = tuple(np.concatenate([m[i] for m in mbs]) for i in range(5))
        ref.train(cat)
    exp = np.concatenate([x.ravel() for x in ref.W])
    # same mathematics, different summation order (per-rank partial sums): fp32 round-off only
    big = np.abs(exp - np.concatenate([x.ravel() for x in xavier_weights(A, 5)])) > 0
    assert np.abs(got - exp)[big].max() < 5e-5
    assert np.mean(np.abs(got - exp) < 1e-6) > 0.99
