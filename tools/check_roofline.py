#!/usr/bin/env python
"""Diffs every committed bench line against the committed rocprofv3 kernel trace of the same round and workload:

    roofline.frac of profiles/rNN_bench20_<workload>.json          (HIP events inside bench.py)
    vs the same figure recomputed from profiles/rNN_kernel_trace_summary_<workload>.txt
       (avg duration of the side's sampler kernels x SURVEY 8(d) flops or bytes per launch, the peak of the line)

A launch = one half-iteration of one side = the kernels `roofline.kernel_per_side` names for it (+ k_pf_prepare where the
product form runs); the line's launch time is the mean over the two sides, so is the trace's.  The two clocks differ by the
packet overheads a trace does not see and by box-to-box spread (the two files come from different processes of one
session): agreement within TOL (15 %) is what is asserted.  Exit code 1 on a mismatch.

    python tools/check_roofline.py [--round r05] [--tol 0.15]
"""
import argparse
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 0.15


def norm(name):
    """'void bpmf::k_sample_pf<64, 3>(bpmf::LrArgs)' / 'k_sample_pf<64,3>' -> 'k_sample_pf<64,3>'"""
    name = name.strip()
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("bpmf::", "")
    m = re.match(r"([A-Za-z_0-9]+(<[^(]*>)?)", name)
    name = m.group(1) if m else name
    name = name.replace(" ", "")
    # trailing default template arguments as the trace spells them
    name = re.sub(r",(double|float)>$", lambda q: ">" if q.group(1) == "float" else ",double>", name)
    return name


def trace_table(path):
    """{normalised kernel name: (calls, avg_us)}; names in the summary are cut at 72 characters"""
    out = {}
    for line in open(path):
        m = re.match(r"^(.{72})\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line.rstrip("\n"))
        if m and not line.startswith("kernel "):
            out.setdefault(norm(m.group(1)), (int(m.group(2)), float(m.group(3))))
    return out


def side_launch_us(side_kernels, table):
    """sum of the avg durations of the kernels one launch of the side consists of"""
    total, used = 0.0, []
    names = [norm(k) for k in side_kernels.split("+")]
    if any(n.startswith("k_sample_pf") for n in names):
        names.append("k_pf_prepare<64>")
    for n in names:
        hit = [k for k in table if k == n or k.startswith(n.rstrip(">")) and n.rstrip(">") + "," in k + ","]
        if n not in table and not hit:
            return None, "kernel %s of the line is not in the trace" % n
        key = n if n in table else hit[0]
        total += table[key][1]; used.append(key)
    return total, used


def check(round_tag, workload, tol=TOL):
    line_f = os.path.join(ROOT, "profiles", "%s_bench20_%s.json" % (round_tag, workload))
    trace_f = os.path.join(ROOT, "profiles", "%s_kernel_trace_summary_%s.txt" % (round_tag, workload))
    if not (os.path.exists(line_f) and os.path.exists(trace_f)):
        return None
    text = [l for l in open(line_f).read().splitlines() if l.startswith("{")]
    j = json.loads(text[-1])
    r = j["strong_10Mx1M"] if workload == "strong_10Mx1M" else j["roofline"]
    table = trace_table(trace_f)
    if workload == "strong_10Mx1M":
        kern = j["strong_10Mx1M"]["kernel"]
        sides = {"movs": kern["items_side"], "users": kern["users_side"]}
        us = {}
        for k, v in sides.items():
            us[k], used = side_launch_us(v, table)
            if us[k] is None:
                return {"workload": workload, "ok": False, "why": used}
        # one kernel name, two launch sizes: the trace average IS the mean over the two sides
        t_us = sum(us.values()) / 2.0
        per_launch = r["algorithmic_bytes_per_launch"]
        frac_trace = per_launch / (t_us * 1e-6) / 1e9 / 8000.0
        frac_line = r["hbm_frac"]
    else:
        sides = r["kernel_per_side"]
        us = {}
        for k, v in sides.items():
            us[k], used = side_launch_us(v, table)
            if us[k] is None:
                return {"workload": workload, "ok": False, "why": used}
        same = len(set(sides.values())) == 1
        t_us = list(us.values())[0] if same else sum(us.values()) / 2.0          # (one kernel for both sides: its avg is already the mean)
        if r["unit"] == "GB/s":
            frac_trace = r["algorithmic_bytes_per_launch"] / (t_us * 1e-6) / 1e9 / r["peak"]
        else:
            frac_trace = r["executed_flops_per_launch"] / (t_us * 1e-6) / 1e12 / r["peak"]
        frac_line = r["frac"]
    rel = abs(frac_trace - frac_line) / max(frac_line, 1e-12)
    return {"workload": workload, "round": round_tag, "frac_line": frac_line, "frac_trace": frac_trace, "launch_us_trace": t_us,
            "launch_us_line": (r.get("launch_ms") or 0.0) * 1e3 if workload != "strong_10Mx1M" else 0.5 * sum(r["sampler_ms"].values()) * 1e3,
            "rel_diff": rel, "ok": rel <= tol}


def newest_round():
    tags = sorted({os.path.basename(f).split("_")[0] for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench20_*.json"))})
    return tags[-1] if tags else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default=None)
    ap.add_argument("--tol", type=float, default=TOL)
    a = ap.parse_args()
    tag = a.round or newest_round()
    if not tag:
        print("no profiles/r*_bench20_*.json")
        return 0
    bad = 0
    print("%-16s %10s %10s %10s %10s %8s" % ("workload (" + tag + ")", "frac line", "frac trace", "us line", "us trace", "diff"))
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_bench20_*.json" % tag))):
        wl = os.path.basename(f)[len(tag) + len("_bench20_"):-len(".json")]
        res = check(tag, wl, a.tol)
        if res is None:
            continue
        if "frac_line" not in res:
            print("%-16s %s" % (wl, res["why"])); bad += 1; continue
        print("%-16s %10.4f %10.4f %10.1f %10.1f %7.1f%%%s" % (wl, res["frac_line"], res["frac_trace"], res["launch_us_line"], res["launch_us_trace"],
                                                              100 * res["rel_diff"], "" if res["ok"] else "   MISMATCH"))
        bad += 0 if res["ok"] else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
