#!/usr/bin/env python3
"""Condense a gpurun_out/prof_rNN tree (scripts/profile_rNN.sh) into the tracked files under profiles/:
  profiles/rNN_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command (verbatim)
  profiles/rNN_pmc.json           FETCH_SIZE / WRITE_SIZE medians per launch, corrected as MI355X_MICROARCH.md
                                  prescribes (FETCH_SIZE x2 on gfx950, verified on the calibration copy) 
  profiles/traffic.json           what bench.py reports as roofline.traffic
usage: summarize_profile.py gpurun_out/prof_r01 r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def counters(d):
    fs = glob.glob(os.path.join(src, d, "*", "*counter_collection.csv"))
    by = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            by[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return by


ks = glob.glob(os.path.join(src, "trace64k", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(out, tag + "_kernel_stats.csv"))
for name in ("trace64k.json", "bench1m.json", "calib_plain.txt"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, tag + "_" + name))

res = {"units": "bytes per launch (median over launches)", "fetch_correction": None, "runs": {}}
cal_f = counters("calib_FETCH_SIZE")
cal_w = counters("calib_WRITE_SIZE")
known = 140 * (1 << 20) * 8
kf = statistics.median(cal_f[("pb::k_calib_copy", "FETCH_SIZE")]) * 1024
kw = statistics.median(cal_w[("pb::k_calib_copy", "WRITE_SIZE")]) * 1024
res["calibration"] = {"kernel": "pb::k_calib_copy (1M filters x 140 components, 8 B/lane buffer loads+stores)",
                      "known_read_bytes": known, "known_write_bytes": known, "FETCH_SIZE_bytes_raw": kf,
                      "WRITE_SIZE_bytes_raw": kw, "fetch_scale": known / kf, "write_scale": known / kw}
fs, ws = known / kf, known / kw
res["fetch_correction"] = "FETCH_SIZE x %.4f, WRITE_SIZE x %.4f (from the calibration copy; the guide's gfx950 rule is x2 / x1)" % (fs, ws)
traffic = {}
for run, B in (("pmc64k", 65536), ("pmc1m", 1 << 20)):
    f = counters(run + "_FETCH_SIZE")
    w = counters(run + "_WRITE_SIZE")
    for (k, c), v in list(f.items()):
        if "k_step" not in k:
            continue
        rd = statistics.median(v) * 1024 * fs
        wr = statistics.median(w[(k, "WRITE_SIZE")]) * 1024 * ws
        alg = 2344 * B
        res["runs"]["%s %s" % (run, k)] = {"batch": B, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                                           "algorithmic_bytes": alg, "traffic_over_algorithmic": (rd + wr) / alg,
                                           "launches": len(v)}
        traffic["k_step<15,true>@%d" % B] = {"hbm_bytes_per_launch": rd + wr, "read": rd, "write": wr, "source": tag}
json.dump(res, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(res["runs"], indent=1))
