"""Copies the summaries of a tools/final_capture.sh run (gpurun_out/cap/) into profiles/rNN_* (tracked, committed).
   python tools/collect_profiles.py r02"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = os.path.join(ROOT, "gpurun_out", "cap")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.strip().startswith("{")]
    return json.loads(lines[-1])


def stats_csv(subdir, out, header):
    f = glob.glob(os.path.join(CAP, subdir, "**", "*kernel_stats.csv"), recursive=True)
    assert f, "no kernel_stats.csv under " + subdir
    f.sort(key=os.path.getmtime)
    with open(os.path.join(P, out), "w") as o:
        o.write("# " + header + "\n")
        o.write(open(f[-1]).read())
    print("wrote", out)


for src, dst in (("bench_default.json", "final_bench.json"), ("bench_short.json", "final_bench_steps20.json"),
                 ("bench_fp16.json", "final_bench_fp16.json"), ("bench_fp16_a6.json", "bench_fp16_a6.json"),
                 ("bench_b256.json", "bench_b256.json"), ("bench_b256_a6.json", "bench_b256_a6.json"),
                 ("bench_b256_fp16.json", "bench_b256_fp16.json"), ("bench_dp1.json", "bench_single_rank_dp.json")):
    try:
        d = last_json_line(os.path.join(CAP, src))
        json.dump(d, open(os.path.join(P, "%s_%s" % (tag, dst)), "w"), indent=1)
        print("wrote %s_%s: %s %s" % (tag, dst, d["value"], d["unit"]))
    except Exception as e:
        print("MISSING", src, repr(e)[:120])

rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
stats_csv("stats", tag + "_final_kernel_stats.csv",
          "rocprofv3 --kernel-trace --stats, bench.py --steps 500 --warmup 100 --no-cpu-baseline --replay-size 100000 (B=32, A=4, fp32), build %s (tools/final_capture.sh); unprofiled full bench of the same build: profiles/%s_final_bench.json" % (rev, tag))
for sub, name, what in (("b256", "b256", "--batch-size 256 --num-actions 3 --steps 200 --warmup 60 (fp32, BASELINE configs[2] shape)"),
                        ("fp16", "fp16", "--datatype float16 --steps 500 --warmup 100 (B=32, A=4)"),
                        ("fp16_b256", "fp16_b256", "--datatype float16 --batch-size 256 --num-actions 3 --steps 200 --warmup 60"),
                        ("bn", "bn", "--batch-norm --steps 300 --warmup 100 (B=32, A=4)")):
    try:
        stats_csv(sub, "%s_other_%s_kernel_stats.csv" % (tag, name), "rocprofv3 --kernel-trace --stats, bench.py %s, build %s (tools/final_capture.sh)" % (what, rev))
    except AssertionError as e:
        print("MISSING", e)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(CAP, "pmc_FETCH_SIZE"), os.path.join(CAP, "pmc_WRITE_SIZE"), os.path.join(P, tag)])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_mfma_util.py"), os.path.join(CAP, "pmc_MfmaUtil"), os.path.join(P, tag)])

# plain-text side measurements of the same capture (agent loop on the synthetic environment, tuple-API loop, generic path)
rows = []
try:
    for l in open(os.path.join(CAP, "agent_loop.log")):
        if "steps_per_second" in l or "num_games" in l or "Epoch" in l or "phase" in l.lower():
            rows.append(l.rstrip())
except Exception as e:
    rows.append("MISSING agent_loop.log " + repr(e)[:80])
try:
    rows.append("--- the same with --synthetic_frame_pool 0 (a new 7 KB random frame generated per environment step, as in rounds 1-3)")
    for l in open(os.path.join(CAP, "agent_loop_pool0.log")):
        if "steps_per_second" in l:
            rows.append(l.rstrip())
except Exception as e:
    rows.append("MISSING agent_loop_pool0.log " + repr(e)[:80])
try:
    rows += ["--- tools/exp/act_stamps.py (one-launch acting forward: phase stamps, predict_state latencies)"] + [l.rstrip() for l in open(os.path.join(CAP, "act_stamps.txt"))][-34:]
except Exception as e:
    rows.append("MISSING act_stamps.txt " + repr(e)[:80])
for f in ("tuple_api_rate.txt", "generic_rate.txt"):
    try:
        rows += ["--- " + f] + [l.rstrip() for l in open(os.path.join(CAP, f)) if "steps/s" in l]
    except Exception as e:
        rows.append("MISSING %s %s" % (f, repr(e)[:80]))
if not os.path.exists(os.path.join(CAP, "agent_loop.log")):
    print("no agent_loop.log in this capture: %s_side_rates.txt left as it is" % tag); sys.exit(0)
open(os.path.join(P, tag + "_side_rates.txt"), "w").write(
    "# tools/final_capture.sh, build %s: python -m simple_dqn_amd.main --replay_size 100000 --random_steps 5000 --train_steps 40000 --test_steps 20000 --epochs 1 (synthetic environment);\n"
    "# tools/exp/tuple_api_rate.py; tools/generic_rate.py\n" % rev + "\n".join(rows) + "\n")
print("wrote %s_side_rates.txt (%d lines)" % (tag, len(rows)))
