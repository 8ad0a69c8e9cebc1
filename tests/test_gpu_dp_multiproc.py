"""the product's data-parallel halves across a PROCESS boundary on the one GPU.

Two processes share device 0 (no RCCL: a 1-GPU box cannot host a 2-rank communicator), each runs libsdqn_hip with
`grad_only`, the flat gradient crosses through torch.distributed gloo, both call apply_update(2B).  Asserted inside the
workers: grad_only applies nothing, replicas (weights + RMSProp state) stay BIT-identical across the processes after every
step.  Asserted here: both ranks measured the same distances to the oracle on the concatenated batch, within the bounds of
the single-process test (tests/test_gpu_parity_r2.py::test_dp_arithmetic_two_learners_one_gpu).  float16: the library's own
half payload (sdqn_net_grad_to_half -> gloo sum in half -> sdqn_net_grad_from_half), no overflow at the default scale.
Also here: the device-side dynamic payload scale (halve per overflow, double after 200 clean steps)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle.dqn_numpy import xavier_weights
from util import make_args, random_minibatch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("datatype", ["float32", "float16"])
def test_two_processes_one_gpu_grad_only_gloo_apply(tmp_path, datatype):
    port, world = _free_port(), 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world), str(port), datatype, str(tmp_path)],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    reps = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(world)]
    assert reps[0]["steps"] == reps[1]["steps"]                      # same gradient, same oracle, same bits on both ranks
    for s, row in enumerate(reps[0]["steps"]):
        for i in range(5):
            emax, frac, fro = row["grad"][i]
            if datatype == "float32":
                # two learners + gloo add accumulate in another order than the oracle's 64-sample sum; a ReLU pre-activation within
                # round-off of 0 may gate one side only (finite, local): >= 98 % within 1e-4 of max|g|, none beyond 2e-3
                assert frac >= 0.98 and emax < 2e-3, ("grad", s, i, emax, frac)
                assert row["weights"][i] < 2e-5, ("weights", s, i, row["weights"][i])
                assert row["state"][i] < 1e-3 + 1e-6, ("state", s, i, row["state"][i])
            else:
                # half activations AND a half payload (2^-11 relative per value): relative Frobenius norm, the bound of the other
                # float16 tests; weights move by lr * g / (sqrt(s) + eps) -> half round-off of g shows up at <= 1e-3 of a step
                assert fro < 5e-2, ("grad", s, i, fro)
                assert row["weights"][i] < 3e-3, ("weights", s, i, row["weights"][i])
    if datatype == "float16":
        assert reps[0]["overflow_steps"] == 0 and reps[0]["payload_state"]["scale_log2"] == 10


def test_fp16_dynamic_payload_scale_state_machine():
    """update_kernel<true>'s device-side scale (state = {flag, log2 scale, clean steps}): seeded at 2^15 with a gradient that
    overflows half there, every apply is skipped (parameters + optimizer state untouched, counter + 1) and halves the scale
    until the payload fits; then training resumes; 200 clean steps double it again."""
    import simple_dqn_amd as sd
    A, B = 4, 32
    net = sd.DeepQNetwork(A, make_args(batch_size=B, datatype="float16"))
    net.set_weights(xavier_weights(A, 11), 1); net.set_weights(xavier_weights(A, 10), 0)
    net.set_option("grad_only", 1)
    net.set_option("dp_half_scale_seed", 15)
    mb = random_minibatch(B, A, 12)
    net.train(mb)
    g = [net.get_layer(i, 3) for i in range(5)]
    gmax = max(float(np.abs(x).max()) for x in g)
    # blow the gradient up so that it overflows half at 2^15 .. 2^13 and fits from 2^12 on: |g| * 2^k > 65504  <=>  k > log2(65504/|g|)
    boost = 65504.0 / gmax / 2.0 ** 12.5
    for i in range(5):
        net.set_layer(i, g[i] * boost, 3)
    w0, s0 = net.get_weights(0), net.get_weights(2)
    seen = []
    for step in range(6):
        for i in range(5):
            net.set_layer(i, g[i] * boost, 3)                         # (from_half rewrites g; start every round from the same sums)
        p = net.grad_to_half()
        net.grad_from_half(p)                                         # 1-rank "all-reduce": the identity on the payload
        st = net.half_payload_state()
        net.apply_update(B)
        after = net.half_payload_state()
        seen.append((st["scale_log2"], st["overflow"], after["scale_log2"], net.overflow_steps()))
        if st["overflow"]:
            assert all(np.array_equal(a, b) for a, b in zip(net.get_weights(0), w0)), step
            assert all(np.array_equal(a, b) for a, b in zip(net.get_weights(2), s0)), step
    # 2^15, 2^14, 2^13 overflow (one skipped step per halving), 2^12 fits and the weights move
    assert [s[0] for s in seen[:4]] == [15, 14, 13, 12], seen
    assert [s[1] for s in seen[:4]] == [1, 1, 1, 0], seen
    assert seen[2][3] == 3 and seen[5][3] == 3, seen
    assert not np.array_equal(net.get_layer(3, 0), w0[3])
    # 200 clean steps double the scale (and the counter restarts)
    small = [x * (boost / 64.0) for x in g]                           # fits at 2^13 as well: the doubled scale must not overflow
    st = net.half_payload_state()
    k0, clean0 = st["scale_log2"], st["clean_steps"]
    for step in range(200 - clean0):
        for i in range(5):
            net.set_layer(i, small[i], 3)
        net.grad_from_half(net.grad_to_half())
        net.apply_update(B)
    st = net.half_payload_state()
    assert st["scale_log2"] == k0 + 1 and st["clean_steps"] == 0 and net.overflow_steps() == 3, st
