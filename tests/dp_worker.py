"""One rank of tests/test_gpu_dp_multiproc.py (not product code).

    python tests/dp_worker.py <rank> <world> <port> <float32|float16> <outdir>

Two PROCESSES share the box's one GPU, each with its own libsdqn_hip handle (its own replica of theta, theta-, s): the
product's two data-parallel halves across a process boundary — `grad_only` step -> exchange of the flat gradient through
torch.distributed gloo (fp32 sum; float16 mode: the library's half payload summed IN half) -> sdqn_net_apply_update(R*B) —
i.e. exactly what sdqn_dp_init + ncclAllReduce do on N GPUs, with gloo standing in for RCCL (a 1-GPU box cannot host a 2-rank
RCCL communicator).  Every rank also runs the oracle on the concatenated R*B batch (deterministic: same bits on both)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, datatype, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    import numpy as np
    import torch                                    # torch first: one HIP runtime per process (simple_dqn_amd/_lib.py)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import simple_dqn_amd as sd
    from oracle.dqn_numpy import OracleDQN, xavier_weights
    from util import make_args, random_minibatch

    A, B, half = 4, 32, datatype == "float16"
    args = make_args(batch_size=B, datatype=datatype, device_id=0)
    net = sd.DeepQNetwork(A, args)
    ws, wt = xavier_weights(A, 651), xavier_weights(A, 652)
    net.set_weights(wt, 1); net.set_weights(ws, 0)
    net.set_option("grad_only", 1)
    o = OracleDQN(A, batch_size=world * B, weights=ws, half_activations=half)
    o.Wt = [w.copy() for w in wt]
    sizes = [net.get_layer(i, 0).size for i in range(5)]
    report = {"steps": [], "overflow_steps": None}
    for s in range(3):
        mbs = [random_minibatch(B, A, 900 + 10 * s + r, reward_range=(-2, 3)) for r in range(world)]
        w_before = net.get_weights(0)
        net.train(mbs[rank])                                            # local gradient SUMS only
        assert all(np.array_equal(a, b) for a, b in zip(net.get_weights(0), w_before)), "grad_only applied an update"
        local = [net.get_layer(i, 3) for i in range(5)]
        if half:
            t = torch.from_numpy(net.grad_to_half())                    # g * 2^k as IEEE half, internal layout
            dist.all_reduce(t, op=dist.ReduceOp.SUM)                    # summed in half, like ncclAllReduce(ncclFloat16)
            net.grad_from_half(t.numpy())
        else:
            flat = torch.from_numpy(np.concatenate([x.ravel() for x in local]))
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)                 # the ONE collective of the step
            off = 0
            for i, n in enumerate(sizes):
                net.set_layer(i, flat[off:off + n].numpy().reshape(local[i].shape), 3); off += n
        gsum = [net.get_layer(i, 3) for i in range(5)]                  # what the optimizer is about to see
        net.apply_update(world * B)
        # ---- the same step on the oracle, concatenated batch (Neon: gradient SUM over the batch, then grad / be.bsz)
        both = tuple(np.concatenate([m[i] for m in mbs]) for i in range(5))
        g, _, _, _ = o.gradients(both)
        o.rmsprop(g, world * B)
        row = {"grad": [], "weights": [], "state": []}
        for i in range(5):
            gmax = max(1e-3, float(np.abs(g[i]).max()))
            err = np.abs(gsum[i] - g[i]) / gmax
            row["grad"].append([float(err.max()), float((err < 1e-4).mean()),
                                float(np.linalg.norm((gsum[i] - g[i]).ravel()) / max(1e-12, np.linalg.norm(g[i].ravel())))])
            big = np.abs(g[i]) / (world * B) > 1e-6
            row["weights"].append(float(np.abs(net.get_layer(i, 0) - o.W[i])[big].max()))
            row["state"].append(float(np.abs(net.get_layer(i, 2) - o.S[i]).max() / (1e-12 + np.abs(o.S[i]).max())))
        report["steps"].append(row)
        # replicas bit-identical across the process boundary (weights AND optimizer state)
        mine = torch.from_numpy(np.concatenate([net.get_layer(i, 0).ravel() for i in range(5)] + [net.get_layer(i, 2).ravel() for i in range(5)]))
        allw = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allw, mine)
        assert all(torch.equal(allw[0], x) for x in allw), "replicas diverged at step %d" % s
        # lock-step for the next round (per-step comparison, not free-running: DESIGN.md §2)
        net.set_weights(o.W, 0); net.set_weights(o.S, 2)
    if half:
        report["overflow_steps"] = net.overflow_steps()
        report["payload_state"] = net.half_payload_state()
    import json
    with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
        json.dump(report, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
