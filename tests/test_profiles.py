"""The committed rocprofv3 evidence must agree with the committed bench line (profiles/README.md): the hot kernel's average
duration from `rocprofv3 --kernel-trace --stats` against `roofline.kernel_avg_us` measured with HIP events by bench.py
WITHOUT the profiler (both come from one scripts/profile.sh run on one box: 8 % covers the tracer and run-to-run spread), and the roofline figures must be recomputable from the line itself.  CPU-only: reads profiles/."""
import csv
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_round():
    tags = sorted({os.path.basename(p)[:3] for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_default.json"))})
    if not tags:
        pytest.skip("no committed bench line")
    return tags[-1]


def bench_line(tag):
    with open(os.path.join(ROOT, "profiles", tag + "_bench_default.json")) as f:
        return json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])


def test_rocprof_average_agrees_with_the_bench_line():
    tag = latest_round()
    d = bench_line(tag)
    with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")) as f:
        rows = list(csv.DictReader(f))
    hot = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    assert "k_step" in hot["Name"]
    prof_us = float(hot["AverageNs"]) / 1e3
    assert abs(prof_us - d["roofline"]["kernel_avg_us"]) / prof_us < 0.08, (prof_us, d["roofline"]["kernel_avg_us"])


def test_roofline_is_recomputable_from_the_line():
    d = bench_line(latest_round())
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    achieved = r["algorithmic_bytes_per_launch"] / (r["kernel_avg_us"] * 1e-6) / 1e9
    assert abs(achieved - r["achieved"]) / achieved < 1e-6 and abs(r["frac"] - achieved / r["peak"]) < 1e-9
    assert r["frac"] < 1.0 and r["frac_cache_busting"] < r["frac"]
    # whole-job throughput cannot beat the kernel's own rate
    batch = r["algorithmic_bytes_per_launch"] // 2344  # 2 344 B per 15-state step
    assert d["value"] <= batch / (r["kernel_avg_us"] * 1e-6) * 1.001
    cb = r["cache_busting"]
    assert abs(cb["algorithmic_bytes_per_launch"] / (cb["kernel_avg_us"] * 1e-6) / 1e9 - cb["achieved"]) / cb["achieved"] < 1e-6
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1


def test_one_million_filter_trace_agrees_with_the_cache_busting_leg():
    """the true-HBM figure (roofline.frac_cache_busting, 1 M filters) has a rocprofv3 kernel trace of its own,
    and its average agrees with the HIP-event average of the bench line's cache-busting leg (same box, same profile run)."""
    tag = latest_round()
    path = os.path.join(ROOT, "profiles", tag + "_kernel_stats_1m.csv")
    if not os.path.exists(path):
        pytest.skip("no 1 M-filter trace for " + tag)
    with open(path) as f:
        rows = list(csv.DictReader(f))
    hot = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    assert "k_step" in hot["Name"]
    prof_us = float(hot["AverageNs"]) / 1e3
    cb = bench_line(tag)["roofline"]["cache_busting"]
    assert cb["batch"] == 1 << 20
    assert abs(prof_us - cb["kernel_avg_us"]) / prof_us < 0.08, (prof_us, cb["kernel_avg_us"])


def test_numbers_quoted_in_design_are_in_the_committed_profiles():
    """every microsecond figure DESIGN.md quotes from a profile must be reproducible from the committed file --
    the table between the quoted-numbers markers of DESIGN.md section 6 names the file, a string that selects one of its lines and the
    value; the first number followed by "us" behind that string must agree within 5 %."""
    import re
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "<!-- quoted-numbers-begin -->" in text and "<!-- quoted-numbers-end -->" in text
    table = text.split("<!-- quoted-numbers-begin -->")[1].split("<!-- quoted-numbers-end -->")[0]
    rows = [r for r in table.splitlines() if r.startswith("| `profiles/")]
    assert len(rows) >= 20
    for r in rows:
        cells = [c.strip() for c in r.strip().strip("|").split("|")]
        # (a cell may itself contain '|' inside its back-ticks: the file is the first cell, the value the last)
        path, want = cells[0].strip("`"), float(cells[-1])
        key = "|".join(cells[1:-1]).strip().strip("`")
        lines = open(os.path.join(ROOT, path), errors="replace").read().splitlines()
        hits = [re.search(re.escape(key) + r"\s*([0-9.]+) us", ln) for ln in lines]
        hits = [h for h in hits if h]
        assert hits, (path, key)
        got = float(hits[0].group(1))
        assert abs(got - want) <= 0.05 * want, (path, key, got, want)


def test_smoother_trace_agrees_with_its_rate_line():
    """The smoother's committed wall-clock rate (scripts/smooth_rate.py, back-to-back launches) and the rocprofv3 kernel trace of the same
    run (scripts/profile.sh) must name the same kernel time: 8 % covers the tracer and the launch gaps."""
    import re
    tag = latest_round()
    txt, stats = os.path.join(ROOT, "profiles", tag + "_smoother.txt"), os.path.join(ROOT, "profiles", tag + "_kernel_stats_smoother.csv")
    if not (os.path.exists(txt) and os.path.exists(stats)):
        pytest.skip("no smoother profile for " + tag)
    rate = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"n=(\d+): \d+ filters, ([\d.]+) us/step", open(txt).read())}
    with open(stats) as f:
        rows = [r for r in csv.DictReader(f) if "k_smooth" in r["Name"]]
    assert rate and rows
    for r in rows:
        n = int(re.search(r"k_smooth_\w+<(\d+)", r["Name"]).group(1))
        assert abs(float(r["AverageNs"]) / 1e3 - rate[n]) / rate[n] < 0.08, (r["Name"], r["AverageNs"], rate[n])
