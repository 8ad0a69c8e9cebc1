"""bench.py's accounting helpers (no GPU): algorithmic work table, roofline entry, committed-profile readers."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


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
    assert e["rocprof_us_per_launch"] and 5 < e["rocprof_us_per_launch"] < 30
    assert 0 < e["frac_of_measured_peak"] < 1 and e["peak_measured"] < e["peak"]
    c = bench.roofline_entry(0, "conv1_fwd(gather+norm+conv+relu)", 0.012, 32, 4)
    assert c["bound"] == "mfma" and c["unit"] == "TFLOP/s"                  # AI ~ 94 FLOP/B: compute-bound in fp32
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
