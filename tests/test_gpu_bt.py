"""The block-tile engine of the throughput regime (B >= 128, float32; simple_dqn_amd/csrc/gemm_engine_bt.h) against the latency
engine it replaces there, stage by stage, and against itself across launch structures.  Parity of the B >= 128 step with the ORACLE is
what tests/test_gpu_dqn.py::test_batch256_one_step, tests/test_gpu_parity_r2.py::test_batch256_* and the B = 160 cases of the conv1
tests assert (they run through this engine by default now); here: the two engines compute the same fp32 sums in different partitions,
so every intermediate must agree to fp32 round-off, ragged blocks included (B = 160: M = 81 B, 49 B ... are not multiples of 64)."""
import numpy as np
import pytest

from oracle.dqn_numpy import OracleDQN, xavier_weights
from util import make_args, random_minibatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    import simple_dqn_amd
    return simple_dqn_amd


def _net(sd, A, B, seed, opts=(), **kw):
    net = sd.DeepQNetwork(A, make_args(batch_size=B, **kw))
    net.set_weights(xavier_weights(A, seed + 1), 1)
    net.set_weights(xavier_weights(A, seed), 0)
    for k, v in opts:
        net.set_option(k, v)
    return net


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


@pytest.mark.parametrize("A,B", [(3, 256), (6, 160), (4, 128)])
def test_block_tile_engine_matches_the_latency_engine(sd, A, B):
    mb = random_minibatch(B, A, 40 + B, reward_range=(-2, 3))
    new = _net(sd, A, B, 7, [("keep_gradients", 1)])
    old = _net(sd, A, B, 7, [("keep_gradients", 1), ("bt", 0)])
    new.train(mb); old.train(mb)
    sizes = dict(a2=2 * B * 81 * 64, a3=2 * B * 49 * 64, a4=2 * B * 512, d4=B * 512, d3p=B * 121 * 64, d2p=B * 121 * 64, d1=B * 400 * 32)
    for name, n in sizes.items():
        x, y = new.debug_read(name, n), old.debug_read(name, n)
        assert _rel(x, y) < 2e-5, (name, _rel(x, y))
    assert np.abs(new.last_q()[0] - old.last_q()[0]).max() < 2e-5
    for i in range(5):
        assert _rel(new.get_layer(i, 3), old.get_layer(i, 3)) < 2e-5, i


@pytest.mark.parametrize("A,B", [(3, 256), (6, 160)])
def test_block_tile_fused_unfused_and_menu_entries_agree(sd, A, B):
    """Fused (bwd3 / bwd2 multi-problem launches) and unfused backward launches use the same block shapes: bit-identical.  Every menu
    entry (other block shapes / prefetch depths: the tuning surface of tools/sweep_bt.py) is the same sum in the same order per output
    element — k ascending, chunk after chunk — so all entries are bit-identical too."""
    mb = random_minibatch(B, A, 50 + B, reward_range=(-2, 3))
    # (fc4 forward / dgrad run on the latency engine by default — too few blocks for this routine to pay there — so the reference of
    #  this comparison selects their block-tile form explicitly: menu entry 1)
    # (conv2 / conv3 forward run on the sample-stationary routine by default since round 6 — another order of the same sums,
    #  test_sample_stationary_forward_convolutions — so the reference selects their former built-in block shape: menu entry 6)
    base = [("keep_gradients", 1), ("bt:3", 1), ("bt:5", 1)]
    ref = _net(sd, A, B, 9, base + [("bt:1", 6), ("bt:2", 6)])
    ref.train(mb)
    g0 = [ref.get_layer(i, 3) for i in range(5)]
    variants = [[("fused_launches", 0), ("bt:1", 6), ("bt:2", 6)]]
    variants += [[("bt:1", m), ("bt:2", m), ("bt:3", m), ("bt:5", m), ("bt:16", min(m, 4)), ("bt:17", min(m, 4))] for m in (1, 2, 3, 4, 5)]
    for opts in variants:
        net = _net(sd, A, B, 9, base + opts)
        net.train(mb)
        assert np.array_equal(net.last_q()[0], ref.last_q()[0]), opts
        for i in range(5):
            assert np.array_equal(net.get_layer(i, 3), g0[i]), (opts, i)


def test_block_tile_fused_rmsprop_step(sd):
    """The fc4_wgrad epilogue of the block-tile engine applies RMSProp in place (12.8 MB of W4 + state never leave as a gradient):
    after one step the weights and the optimizer state equal the materialised-gradient path's (same per-element operations)."""
    A, B = 3, 256
    mb = random_minibatch(B, A, 77)
    fused = _net(sd, A, B, 11)
    split = _net(sd, A, B, 11, [("keep_gradients", 1)])
    fused.train(mb); split.train(mb)
    for which in (0, 2):
        for i in range(5):
            assert np.array_equal(fused.get_layer(i, which), split.get_layer(i, which)), (which, i)


@pytest.mark.parametrize("A,B", [(3, 256), (6, 160), (4, 136)])
def test_conv1_weight_gradient_block_tile_bf16(sd, A, B):
    """conv1's weight gradient at B >= 128: bytes x three exact bf16 planes of delta1 on packed-bf16 MFMA, one workgroup per K slab
    (c1w_bt_kernel).  Every product is exact and the accumulation fp32, so gW1 agrees with the fp32-MFMA engine's to fp32 round-off —
    from the staged minibatch and from the ring (fused gather), ragged slabs and a batch that is not a multiple of 32 included — and
    with other slab sizes."""
    from oracle.replay_numpy import synthetic_fill
    mb = random_minibatch(B, A, 60 + B, reward_range=(-2, 3))
    new = _net(sd, A, B, 21, [("keep_gradients", 1)])
    old = _net(sd, A, B, 21, [("keep_gradients", 1), ("bt:18", -1), ("bt:11", -1)])
    alt = _net(sd, A, B, 21, [("keep_gradients", 1), ("tps:1", 25)])      # 800 positions per slab: whole 80-position chunks too (5 | 25)
    unf = _net(sd, A, B, 21, [("keep_gradients", 1), ("fused_launches", 0)])
    for n in (new, old, alt, unf):
        n.train(mb)
    g_old = old.get_layer(0, 3)
    assert _rel(new.get_layer(0, 3), g_old) < 2e-6 and _rel(alt.get_layer(0, 3), g_old) < 2e-6
    assert np.array_equal(unf.get_layer(0, 3), new.get_layer(0, 3))
    # the ring path (indexes in device memory, frames gathered from the HBM mirror)
    args = make_args(batch_size=B)
    mem = sd.ReplayMemory(3000, args)
    synthetic_fill(mem, 22, num_actions=A)
    idx = np.random.RandomState(23).randint(10, 2900, size=B).astype(np.int64)
    new.train_indexes(mem, idx); old.train_indexes(mem, idx)
    assert _rel(new.get_layer(0, 3), old.get_layer(0, 3)) < 2e-6


def _close_but_for_gate_flips(x, y, tol, what, frac=1e-5):
    """fp32-class agreement of two buffers downstream of ReLU gates: an activation that is +-1e-8 on the two sides flips its gate and
    moves the delta elements behind it by their full size (one flipped conv2 gate reaches up to 4 x 4 x 32 elements of delta1) — all
    other elements within tol of max|y|."""
    d = np.abs(x - y) / max(1e-6, float(np.abs(y).max()))
    assert float((d > tol).mean()) < frac, (what, float((d > tol).mean()), float(d.max()))


def test_block_tile_other_slab_counts(sd):
    """K-slab choices of the weight gradients and of fc4 forward (options tps:<l>, s4) only regroup the fp32 sums."""
    A, B = 3, 256
    mb = random_minibatch(B, A, 78)
    ref = _net(sd, A, B, 13, [("keep_gradients", 1)])
    ref.train(mb)
    for opts in ([("tps:2", 8), ("tps:3", 7)], [("s4", 1)], [("s4", 4), ("tps:2", 41)]):
        net = _net(sd, A, B, 13, [("keep_gradients", 1)] + opts)
        net.train(mb)
        assert np.abs(net.last_q()[0] - ref.last_q()[0]).max() < 2e-5, opts
        for i in range(5):
            assert _rel(net.get_layer(i, 3), ref.get_layer(i, 3)) < 2e-5, (opts, i)


@pytest.mark.parametrize("A,B", [(3, 256), (6, 160)])
def test_xcd_contiguous_block_maps_are_placement_only(sd, A, B):
    """Round 4: fc4_dgrad / bwd3 / bwd2 of the B >= 128 float32 step run on XCD-contiguous block maps (option bt_xcd, on by default): which
    workgroup computes which block changes, nothing else — every gradient, delta and Q-value is bit-identical with the round-robin maps,
    ragged grids included (B = 160: block counts that are not multiples of 8)."""
    mb = random_minibatch(B, A, 70 + B, reward_range=(-2, 3))
    on = _net(sd, A, B, 11, [("keep_gradients", 1)])
    off = _net(sd, A, B, 11, [("keep_gradients", 1), ("bt_xcd", 0)])
    on.train(mb); off.train(mb)
    for name, n in dict(d3p=B * 121 * 64, d2p=B * 121 * 64, d1=B * 400 * 32).items():
        assert np.array_equal(on.debug_read(name, n), off.debug_read(name, n)), name
    for i in range(5):
        assert np.array_equal(on.get_layer(i, 3), off.get_layer(i, 3)), i
    assert np.array_equal(on.last_q()[0], off.last_q()[0])



@pytest.mark.parametrize("A,B", [(3, 256), (4, 160), (6, 136)])
def test_conv1_forward_forms_at_large_batch(sd, A, B):
    """conv1 forward at B >= 128 is a persistent row-chunk pipeline with specialised waves (conv1_bf16_rows2_kernel: 10 matrix waves + 4
    staging waves, v_mfma_f32_16x16x32_bf16) — against the latency regime's per-tile kernel (bt:0 = -1, 32 x 32 x 16 tiles): the same
    exact products (byte x bf16 plane) added in another order, 1e-6 of max|a1|, for both nets, from the staged minibatch and from the
    ring (fused gather), ragged workgroup loads included (B = 136: 68 workgroups per net take two samples each, B = 160: 80); and
    bit-stable from net to net.  (Rounds 4-5's two other forms, bit-identical / 1e-6 to this one while they lived, left the tree in
    round 6: tools/exp/conv1_forms_r5.hip.txt.)"""
    import ctypes as C
    from bench import fill_ring
    mb = random_minibatch(B, A, 911)
    args = make_args(batch_size=B)
    mem = sd.ReplayMemory(3000, args)
    fill_ring(mem, 912, A)
    outs = {}
    for key, form in (("new", 0), ("again", 0), ("ref", -1)):
        net = _net(sd, A, B, 910, opts=(("bt:0", form),))
        net.set_option("grad_only", 1)
        net.train(mb)                                                     # staged host minibatch: both nets' conv1 (z = 0 online, 1 target)
        host = net.debug_read("a1", 2 * B * 400 * 32).copy()
        mt = (C.c_uint32 * 625)(); sd.load().sdqn_mt_seed(mt, 913)
        net.train_from_memory(mem, 1, mt_state=mt, want_cost=False)       # ring path: the gather fused into the kernel's loads
        outs[key] = (host, net.debug_read("a1", 2 * B * 400 * 32).copy())
    for k in (0, 1):
        assert np.array_equal(outs["new"][k], outs["again"][k]) and np.abs(outs["new"][k]).max() > 0
        assert np.abs(outs["new"][k] - outs["ref"][k]).max() <= 1e-6 * np.abs(outs["ref"][k]).max()


@pytest.mark.parametrize("A,B", [(3, 256), (6, 160), (4, 136), (4, 128), (3, 129)])
def test_sample_stationary_forward_convolutions(sd, A, B):
    """Round 6: conv2 / conv3 forward at B >= 128 on the sample-stationary routine (csrc/conv_ss.h: whole samples staged once in LDS,
    im2col at ds_read time, weights streamed through an LDS ring, exact-fp32 16 x 16 x 4 MFMA, specialised staging waves).  Same
    products, one accumulator per output, another k order than the block-tile routine (menu entry 6, the former default): every
    activation agrees to fp32 round-off — two samples per workgroup (2 B > 256), one (B = 128), an odd batch whose last workgroup
    holds one sample (B = 129) — and the result is bit-stable from run to run, from net to net and between the chained (one launch for
    both layers) and the unchained (menu entry 8) form.  (By default the routine runs where
    its workgroups fill at least 80 % of whole rounds of the chip — B = 128, B >= 208 — and leaves the sizes in between to the
    block-tile engine; menu entry 7 selects it whatever the batch size.)"""
    mb = random_minibatch(B, A, 300 + B, reward_range=(-2, 3))
    force = [] if B in (128, 256) else [("bt:1", 7), ("bt:2", 7)]
    new = _net(sd, A, B, 31, [("keep_gradients", 1)] + force)
    # when both layers run on the routine they are ONE launch (conv_ss_chain_kernel: a workgroup's conv3 follows its own conv2 behind a
    # barrier, its input image written from the staging threads' registers): menu entry 8 = the same routine as two launches — same bits
    again = _net(sd, A, B, 31, [("keep_gradients", 1), ("bt:1", 8), ("bt:2", 8)])
    old = _net(sd, A, B, 31, [("keep_gradients", 1), ("bt:1", 6), ("bt:2", 6)])
    # predict() (one net: the routine runs with one sample per workgroup at B = 256) on the same weights
    q1, q2 = new.predict(mb[0]), old.predict(mb[0])
    assert np.abs(q1 - q2).max() < 2e-6 and np.abs(q1 - again.predict(mb[0])).max() < 2e-6      # (one net at B = 128 fills half the chip: the default declines)
    for n in (new, again, old):
        n.train(mb)
    for name, cnt in dict(a2=2 * B * 81 * 64, a3=2 * B * 49 * 64, a4=2 * B * 512).items():
        x, y = new.debug_read(name, cnt), old.debug_read(name, cnt)
        assert _rel(x, y) < 2e-6, (name, _rel(x, y))
        assert np.array_equal(x, again.debug_read(name, cnt)), name
    assert np.abs(new.last_q()[0] - old.last_q()[0]).max() < 2e-6
    for i in range(5):
        assert np.array_equal(new.get_layer(i, 3), again.get_layer(i, 3)), i
    new.train(mb); again.train(mb); old.train(mb)      # a second step from the updated weights: still the same bits
    assert np.array_equal(new.debug_read("a3", 2 * B * 49 * 64), again.debug_read("a3", 2 * B * 49 * 64))


@pytest.mark.parametrize("A,B", [(3, 256), (6, 160), (4, 128), (3, 129), (4, 255)])
def test_float16_conv2_conv3_chains_forward_and_backward(sd, A, B):
    """Round 6, float16 mode at B >= 128 (csrc/conv_ssh.h).  FORWARD: conv2 -> conv3 as ONE launch (a workgroup's samples, W2 and —
    through registers — W3 fetched once, im2col at ds_read time on v_mfma_f32_16x16x32_f16, the conv3 image written from the
    accumulators, a2 / a3 stored as whole lines).  BACKWARD: conv3_dgrad -> conv2_dgrad as ONE launch (one workgroup per sample: the padded
    delta3 plane, W3 and the gate a2 in LDS, delta2 written gated into a second padded LDS plane that the four parity classes of the
    stride-2 transposed convolution read; delta1 collected as a dense plane and gated by a1 on the way out).  Same half operands, fp32
    accumulation in the same k order in one accumulator per output as the packed-fp16 block-tile routines (menu entry 6): Q-values and
    every gradient are BIT-IDENTICAL — two samples per forward workgroup (B = 256, 255, 160, 129), one (B = 128; predict at any size), odd
    batches (129, 255) — with write-through (7) and plain (8) output stores, over two steps."""
    mb = random_minibatch(B, A, 400 + B, reward_range=(-2, 3))
    ids = (1, 2, 7, 9)                                             # conv2_fwd, conv3_fwd, conv3_dgrad, conv2_dgrad
    mk = lambda menu: _net(sd, A, B, 41, [("keep_gradients", 1)] + ([("bt:%d" % i, menu) for i in ids] if menu is not None else []), datatype="float16")
    nets = {"default": mk(None), "forced": mk(7), "plain": mk(8), "bt": mk(6)}
    qs = {k: n.predict(mb[0]).copy() for k, n in nets.items()}
    for k in ("default", "forced", "plain"):
        assert np.array_equal(qs[k], qs["bt"]) and np.abs(qs[k]).max() > 0, k
    for step in range(2):
        for n in nets.values():
            n.train(mb)
        for k in ("default", "forced", "plain"):
            assert np.array_equal(nets[k].last_q()[0], nets["bt"].last_q()[0]), (k, step)
            for i in range(5):
                assert np.array_equal(nets[k].get_layer(i, 3), nets["bt"].get_layer(i, 3)), (k, step, i)
                assert np.abs(nets[k].get_layer(i, 3)).max() > 0
    # the launch structure says so: in float16 mode the chained launches are the default at every batch size
    counts = {}
    for k in ("default", "bt"):
        n = nets[k]
        n.profile(True, -1); n.profile_reset()
        for _ in range(3):
            n.train(mb)
        counts[k] = {p["name"].split("(")[0]: p["launches"] for p in n.profile_read() if p["launches"] > 0}
        n.profile(False)
    assert counts["bt"].get("conv3_fwd", 0) == 3 and counts["bt"].get("conv2_dgrad", 0) == 3 and counts["bt"].get("conv1_fwd", 0) == 3
    assert counts["default"].get("conv3_fwd", 0) == 0 and counts["default"].get("conv2_dgrad", 0) == 0, counts["default"]
    assert counts["default"]["conv2_fwd"] == 3 and counts["default"]["conv3_dgrad"] == 3
    # ... and conv1 rides in front of the forward chain (the workgroup computes its own samples' a1 from the ring's frames): no launch of its own
    assert counts["default"].get("conv1_fwd", 0) == 0, counts["default"]


@pytest.mark.parametrize("A,B", [(4, 32), (6, 32), (3, 64), (4, 100), (3, 5), (4, 1)])
def test_float16_forward_chain_below_the_throughput_regime(sd, A, B):
    """float16, B < 128: conv1 -> conv2 -> conv3 forward is the same one launch (2 B workgroups of one sample; conv2 + conv3 alone: +7 %
    steps/s at B = 32, +15 % at B = 100) in place of the latency engine's three K-split launches (menu entry 6 -> declined -> those): same half operands, another order
    of the fp32 sums — Q-values agree to fp32 round-off of half-rounded activations, both are held to the half oracle, run-to-run stable."""
    from oracle.dqn_numpy import OracleDQN
    mb = random_minibatch(B, A, 500 + B, reward_range=(-2, 3))
    new = _net(sd, A, B, 51, datatype="float16")
    old = _net(sd, A, B, 51, [("bt:1", 6), ("bt:2", 6)], datatype="float16")
    o = OracleDQN(A, batch_size=B, weights=xavier_weights(A, 51), half_activations=True)
    q1, q2, qo = new.predict(mb[0]).copy(), old.predict(mb[0]).copy(), o.predict(mb[0])
    assert np.array_equal(q1, new.predict(mb[0]))
    assert np.abs(q1 - q2).max() < 1e-3 and np.abs(q1 - qo).max() < 3e-3 and np.abs(q2 - qo).max() < 3e-3     # (below B = 48 the two also differ in conv1's input semantics: 2e-4)
    for n in (new, old):
        n.profile(True, -1); n.profile_reset()
        for _ in range(3):
            n.train(mb)
    cn = {p["name"].split("(")[0]: p["launches"] for p in new.profile_read() if p["launches"] > 0}
    co = {p["name"].split("(")[0]: p["launches"] for p in old.profile_read() if p["launches"] > 0}
    assert cn.get("conv3_fwd", 0) == 0 and cn.get("conv1_fwd", 0) == 0 and cn["conv2_fwd"] == 3 and co["conv3_fwd"] == 3 and co["conv1_fwd"] == 3, (cn, co)
    # (no free-running comparison: three half-precision steps from two summation orders drift 1e-3 .. 3e-2 apart with |Q| ~ 1)


@pytest.mark.parametrize("A,B", [(3, 256), (4, 129), (6, 160)])
def test_float16_conv1_weight_gradient_inside_the_weight_gradient_launch(sd, A, B):
    """Round 6, float16 at B >= 128: conv1's weight gradient (c1w_h_kernel's K-slab workgroups) is a fourth block-id range of the launch that
    carries fc4_wgrad (+ RMSProp) || conv3_wgrad || conv2_wgrad — delta1 is complete before it starts — instead of a launch of its own
    (option c1w_in_wgrads: 0 = own launch, 1 = last in the block order (built-in), 2 = first).  Same body, same slabs: every gradient and the
    updated network are bit-identical over two steps; the launch structure shows one launch fewer."""
    mb = random_minibatch(B, A, 600 + B, reward_range=(-2, 3))
    nets = {}
    for m in (0, 1, 2):
        n = _net(sd, A, B, 61, [("keep_gradients", 1), ("c1w_in_wgrads", m)], datatype="float16")
        n.train(mb); n.train(mb)
        nets[m] = n
    for m in (1, 2):
        for i in range(5):
            assert np.array_equal(nets[m].get_layer(i, 3), nets[0].get_layer(i, 3)) and np.abs(nets[m].get_layer(i, 3)).max() > 0, (m, i)
        assert np.array_equal(nets[m].predict(mb[0]), nets[0].predict(mb[0]))
    counts = {}
    for m in (0, 1):
        n = nets[m]
        n.profile(True, -1); n.profile_reset()
        for _ in range(3):
            n.train(mb)
        counts[m] = {p["name"].split("(")[0]: p["launches"] for p in n.profile_read() if p["launches"] > 0}
        n.profile(False)
    assert counts[0].get("bwd1", 0) == 3 and counts[1].get("bwd1", 0) == 0 and counts[1]["wgrads"] == 3, counts
