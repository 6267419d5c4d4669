"""PATCH.py argument of tools/build_variant.sh for the patches in this directory:
    BPMF_PATCH=tools/patches/wg2_gram44_f64.patch tools/build_variant.sh g44 tools/patches/apply.py
Every patch names the revision it was made against in its first line (`# base: <commit>`): the kernels moved on since (round 6
changed kernels_wg2.h / kernels_slab.h again), so the COPY of csrc/ that build_variant.sh builds from is first replaced by
that revision's sources (git archive), then patched -- the documented A/B measurements of docs/FINDINGS.md 17-19 stay
reproducible from the tree.  tests/test_profiles.py::test_experiment_patches_apply dry-runs every patch the same way."""
import os, re, subprocess, sys
csrc = sys.argv[1]                                   # <copy>/bpmf_amd/csrc
root = os.path.dirname(os.path.dirname(csrc))
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in os.environ["BPMF_PATCH"].split(":"):
    p = os.path.abspath(p)
    m = re.match(r"# base:\s*([0-9a-f]+)", open(p).readline())
    if m:
        tar = subprocess.run(["git", "-C", repo, "archive", m.group(1), "bpmf_amd/csrc", "include"], capture_output=True, check=True).stdout
        subprocess.run(["rm", "-rf", csrc]); os.makedirs(csrc)
        subprocess.run(["tar", "-x", "-C", root], input=tar, check=True)
    subprocess.check_call(["patch", "-p1", "-d", root, "-i", p])
