#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (scripts/profile.sh) into the tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv, <tag>_kernel_stats_n21.csv   rocprofv3 --kernel-trace --stats summaries (verbatim)
  profiles/<tag>_pmc.json    FETCH_SIZE / WRITE_SIZE medians per launch, scaled by the calibration copy
                             (MI355X_MICROARCH.md section HBM: FETCH_SIZE x2 on gfx950; verified here, not assumed)
  profiles/traffic.json      what bench.py reports as roofline.traffic (keyed "<kernel>@<batch>")
usage: profile_summary.py <tag>
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import statistics
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def kname(raw):
    """rocprofv3's demangled name -> what pb_hot_kernel() reports: k_step_coop<15,true,1> for the plain fused step
    (no second measurement, with predict); the other instantiations keep their full argument list."""
    n = re.sub(r"\s+", "", raw.split("(")[0].replace("void ", "").replace("pb::", ""))
    return re.sub(r"^(k_step_coop<\d+,true,\d),Corr<false>,true>$", r"\1>", n)


def newest(pattern):
    """gpurun MERGES a run's output into gpurun_out/: a directory can hold the files of several runs (one pid prefix
    each); only the newest one counts."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


def counters(d):
    by = collections.defaultdict(list)
    for f in newest(os.path.join(src, d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = kname(r["Kernel_Name"])
            by[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return by


for d, name in (("trace64k", "_kernel_stats.csv"), ("trace64k_n21", "_kernel_stats_n21.csv"), ("trace1m", "_kernel_stats_1m.csv"),
                ("shim/trace", "_kernel_stats_shim.csv"), ("shim/trace21", "_kernel_stats_shim_n21.csv"),
                ("others", "_kernel_stats_others.csv"), ("configs", "_kernel_stats_configs.csv"),
                ("smoother", "_kernel_stats_smoother.csv")):
    ks = newest(os.path.join(src, d, "*", "*kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(out, tag + name))
for name in ("trace64k.json", "trace64k_n21.json", "trace1m.json", "leg_rates.txt", "bench1m.json", "calib_plain.txt", "copybench.txt", "batch_sweep.txt",
             "bench_default.json", "others.txt", "configs.txt", "smoother.txt", "n21_input_footprint.txt", "checkpoint_rate.txt",
             "smoother_pivoted.txt", "smoother_reg.txt", "smoother_lane.txt", "leg_ab.txt", "segment_rate.txt", "smooth_log.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, tag + "_" + name))

p = os.path.join(src, "shim", "shim_sweep.txt")
if os.path.exists(p):
    with open(p, errors="replace") as f, open(os.path.join(out, tag + "_shim_sweep.txt"), "w") as o:
        o.writelines(l for l in f if l.startswith("shim sweep"))   # (the file also holds the handlers' chatter on stdout / stderr)
# per-kernel MEDIAN durations of the handler-path traces: the stats file's average includes the first launch (module load,
# tens of milliseconds), the median does not
for d in ("shim/trace", "shim/trace21"):
    for f in newest(os.path.join(src, d, "*", "*kernel_trace.csv")):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            by[kname(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        with open(os.path.join(out, tag + "_kernel_medians_" + d.split("/")[1].replace("trace", "shim") + ".txt"), "w") as o:
            o.write("# median / mean-without-the-first-launch duration per kernel (us), from rocprofv3 --kernel-trace of scripts/shim_rate.sh\n")
            for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                rest = v[1:] if len(v) > 1 else v
                o.write("%-60s calls %6d  median %9.2f  mean w/o first %9.2f  first %12.2f\n" % (k[:60], len(v), statistics.median(v), sum(rest) / len(rest), v[0]))

known = 140 * (1 << 20) * 8
kf = statistics.median(counters("calib_FETCH_SIZE")[("k_calib_copy", "FETCH_SIZE")]) * 1024
kw = statistics.median(counters("calib_WRITE_SIZE")[("k_calib_copy", "WRITE_SIZE")]) * 1024
fs, ws = known / kf, known / kw
res = {"units": "bytes per launch (median over launches)",
       "calibration": {"kernel": "k_calib_copy (1M filters x 70 rows of a tile, 16 B/lane buffer loads+stores)",
                       "known_read_bytes": known, "known_write_bytes": known, "FETCH_SIZE_bytes_raw": kf,
                       "WRITE_SIZE_bytes_raw": kw, "fetch_scale": fs, "write_scale": ws},
       "runs": {}}
traffic = {}
for run, B, bps in (("pmc64k", 65536, 2344), ("pmc1m", 1 << 20, 2344), ("pmc64k_n21", 65536, 4216)):
    f, w = counters(run + "_FETCH_SIZE"), counters(run + "_WRITE_SIZE")
    for (k, c), v in list(f.items()):
        if not k.startswith("k_step"):
            continue
        rd = statistics.median(v) * 1024 * fs
        wr = statistics.median(w[(k, "WRITE_SIZE")]) * 1024 * ws
        res["runs"]["%s %s" % (run, k)] = {"batch": B, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                                           "algorithmic_bytes": bps * B, "traffic_over_algorithmic": (rd + wr) / (bps * B),
                                           "launches": len(v)}
        traffic["%s@%d" % (k, B)] = {"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr, "source": tag}
json.dump(res, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
# the smoother's SQ / LDS counters (two passes), averaged per launch
sm = {}
for d in ("smooth_pmc_a", "smooth_pmc_b", "smooth_pmc_fetch", "smooth_pmc_write"):
    if not glob.glob(os.path.join(src, d, "*", "*counter_collection.csv")):
        continue
    for (k, c), v in counters(d).items():
        m = re.match(r"k_smooth_(?:wide|lane|reg)<(\d+)", k)   # k_smooth_wide<15> (default for 15 states since round 5), k_smooth_lane<NS>, k_smooth_reg<NS,PIVOT>
        if m:
            sm.setdefault("n" + m.group(1), {})[c] = sum(v) / len(v)
            sm["n" + m.group(1)]["kernel"] = k.split("(")[0]
for r in sm.values():  # HBM traffic with the calibration of the hot step's counters (same units, same corrections)
    if "FETCH_SIZE" in r:
        r["hbm_read_bytes"] = r["FETCH_SIZE"] * 1024 * fs
    if "WRITE_SIZE" in r:
        r["hbm_write_bytes"] = r["WRITE_SIZE"] * 1024 * ws
txt = os.path.join(src, "smoother.txt")
if sm:
    for line in open(txt) if os.path.exists(txt) else ():
        m = re.search(r"n=(\d+): \d+ filters, ([\d.]+) us/step", line)
        if m and "n" + m.group(1) in sm:
            sm["n" + m.group(1)]["launch_us"] = float(m.group(2))
    for r in sm.values():
        # SQ_LDS_IDX_ACTIVE: cycles the LDS index pipe is busy, summed over the 256 CUs; kernel cycles at 2.4 GHz
        if "SQ_LDS_IDX_ACTIVE" in r and "launch_us" in r:
            r["lds_busy_fraction_per_cu"] = r["SQ_LDS_IDX_ACTIVE"] / 256 / (r["launch_us"] * 2400.0)
        if "SQ_LDS_BANK_CONFLICT" in r and r.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_bank_conflict_share"] = r["SQ_LDS_BANK_CONFLICT"] / r["SQ_LDS_IDX_ACTIVE"]
        if r.get("SQ_WAVES"):
            r["valu_insts_per_wave"] = r.get("SQ_INSTS_VALU", 0) / r["SQ_WAVES"]
            r["lds_insts_per_wave"] = r.get("SQ_INSTS_LDS", 0) / r["SQ_WAVES"]
    json.dump({"what": "rocprofv3 --pmc of scripts/smooth_rate.py (64k filters), two passes (scripts/profile.sh); averages per launch",
               "kernel": "pb_smooth_step's default kernel (15 states: k_smooth_wide<15> since the second half of round 5; 21 states: k_smooth_lane<21>; PRONTO_SMOOTH_KERNEL=lane / reg: the older kernels)", "runs": sm},
              open(os.path.join(out, tag + "_smoother_pmc.json"), "w"), indent=1)

for k, v in res["runs"].items():
    print("%-40s read %.1f MB write %.1f MB = %.3f x algorithmic" % (k, v["hbm_read_bytes"] / 1e6, v["hbm_write_bytes"] / 1e6, v["traffic_over_algorithmic"]))
for f in sorted(glob.glob(os.path.join(out, tag + "_kernel_stats*.csv"))):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Name"]:
            print(os.path.basename(f), r["Name"].split("(")[0], "calls", r["Calls"], "avg_ns", r["AverageNs"])
