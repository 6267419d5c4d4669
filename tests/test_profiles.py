"""The committed bench lines and the committed rocprofv3 traces of the newest round must tell the same story:
tools/check_roofline.py recomputes every workload's `roofline.frac` from the trace summary (avg kernel duration x SURVEY 8(d)
flops / bytes per launch) and diffs it with the figure bench.py printed from its own HIP events."""
import importlib.util
import os

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
