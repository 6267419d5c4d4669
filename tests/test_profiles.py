"""The committed bench lines and the committed rocprofv3 traces of the newest round must tell the same story:
tools/check_roofline.py recomputes every workload's `roofline.frac` from the trace summary (avg kernel duration x SURVEY 8(d)
flops / bytes per launch) and diffs it with the figure bench.py printed from its own HIP events."""
import importlib.util
import os

import pytest

from tests.conftest import ROOT


def _mod():
    spec = importlib.util.spec_from_file_location("check_roofline", os.path.join(ROOT, "tools", "check_roofline.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_lines_agree_with_the_committed_traces():
    m = _mod()
    tag = m.newest_round()
    assert tag is not None
    seen = 0
    for wl in ("ml1m", "ml1m_k64", "chembl", "ml1m_k128", "ml1m_k128_f64", "strong_10Mx1M"):
        res = m.check(tag, wl)
        if res is None:
            continue
        seen += 1
        assert res.get("ok"), res
    assert seen >= 1, "no bench20 line + kernel trace pair under profiles/ for round %s" % tag


def test_kernel_name_normalisation():
    m = _mod()
    assert m.norm("void bpmf::k_sample1<32>(bpmf::SampleArgs, bpmf::FusedArgs)") == "k_sample1<32>"
    assert m.norm("void bpmf::k_sample_pf<64, 3>(bpmf::LrArgs)") == "k_sample_pf<64,3>"
    assert m.norm("k_sample_wg2<128,4,double>") == "k_sample_wg2<128,4,double>"
    assert m.norm("void bpmf::k_sample_wg2<128, 2, float>(bpmf::SampleArgs, bpmf::StatRiders)") == "k_sample_wg2<128,2>"


def test_experiment_patches_apply(tmp_path):
    """tools/patches/*.patch (the kernel variants behind docs/FINDINGS.md 17-19) must apply to the revision each one names in its
    `# base:` line -- what tools/patches/apply.py checks out before patching (two of the three had stopped applying
    to HEAD and nothing noticed).  Needs the git history; skipped in an export without it."""
    import glob
    import re
    import subprocess
    patches = sorted(glob.glob(os.path.join(ROOT, "tools", "patches", "*.patch")))
    assert patches
    if subprocess.run(["git", "-C", ROOT, "rev-parse", "--git-dir"], capture_output=True).returncode != 0:
        pytest.skip("no git history here")
    for p in patches:
        m = re.match(r"# base:\s*([0-9a-f]+)", open(p).readline())
        assert m, "%s does not name its base revision" % p
        if subprocess.run(["git", "-C", ROOT, "cat-file", "-e", m.group(1) + "^{commit}"], capture_output=True).returncode != 0:
            pytest.skip("base revision %s is not in this clone" % m.group(1))
        d = tmp_path / os.path.basename(p)
        d.mkdir()
        tar = subprocess.run(["git", "-C", ROOT, "archive", m.group(1), "bpmf_amd/csrc", "include"], capture_output=True, check=True).stdout
        subprocess.run(["tar", "-x", "-C", str(d)], input=tar, check=True)
        r = subprocess.run(["patch", "-p1", "--dry-run", "-d", str(d), "-i", p], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stdout[-600:])


def test_committed_line_carries_every_single_gpu_config():
    """The newest committed driver-style line (profiles/rNN_bench_20steps.json: `python bench.py --gpus 1 --steps 20 --warmup 5`) must be
    what DESIGN.md section 5 says it is: the headline + the four other single-GPU BASELINE configurations as `configs` legs, each with a
    roofline object and a GREEN chain parity against the oracle, a timed window of >= 2 s, `issue_bound` / `mfma_shape` in the
    roofline, and the strong-scaling record saying what its traffic counter is."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_20steps.json")))
    assert files
    tag = os.path.basename(files[-1]).split("_")[0]
    if tag < "r06":
        pytest.skip("lines before round 6 have no `configs` object")
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    assert d["steps"] == 20 and d["n_gpus"] == 1 and d["value"] > 5e7 and d["timed_window_s"] >= 2.0 and d["repeats"] >= 100
    assert d["parity"]["ok"] and d["parity"]["d_rmse_max"] < 1e-6
    assert d["handover"]["ms"] > 0 and d["handover"]["value_incl_handover"]["1000"] < d["value"]      # (PCIe-inclusive rate: DESIGN.md section 5)
    rf = d["roofline"]
    assert 0.5 < rf["issue_bound"]["frac_of_launch"] < 1.0 and rf["issue_bound"]["current"]
    assert rf["mfma_shape"]["frac_of_shape_peak"] > rf["frac"]
    assert set(d["configs"]) == {"chembl", "ml1m_k128", "ml1m_k64", "ml1m_k128_f64"}
    for name, c in d["configs"].items():
        assert c.get("value"), (name, c.get("error"))
        assert c["steps"] == 20 and c["timed_window_s"] >= 2.0 and c["parity"]["ok"], name
        assert 0.0 < c["roofline"]["frac"] < 1.0 and set(c["launch_us_per_side"]) == {"movs", "users"}, name
    assert d["configs"]["ml1m_k128"]["dtype"] == "f32" and d["configs"]["ml1m_k128_f64"]["dtype"] == "f64"
    s = d["strong_10Mx1M"]
    assert "Infinity-Cache" in s["traffic_is"] and set(s["per_side_counters"]) == {"items_side", "users_side"}
    assert 0.8 < s["hbm_frac"] < 1.0
