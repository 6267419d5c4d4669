"""PATCH.py argument of tools/build_variant.sh for the patches in this directory:
    BPMF_PATCH=tools/patches/wg2_lookahead.patch tools/build_variant.sh la tools/patches/apply.py
applies the patch (made with `git diff` at the repo root) to the COPY of csrc/ that build_variant.sh builds from."""
import os, subprocess, sys
csrc = sys.argv[1]                                   # <copy>/bpmf_amd/csrc
root = os.path.dirname(os.path.dirname(csrc))
for p in os.environ["BPMF_PATCH"].split(":"):
    subprocess.check_call(["patch", "-p1", "-d", root, "-i", os.path.abspath(p)])
