"""PATCH.py argument of tools/build_variant.sh: compile-time switches for a variant build --
    BPMF_DEFINES="BPMF_SLAB_DEPTH=8 BPMF_SLAB_WPS=3" tools/build_variant.sh d8 tools/patches/define.py
adds -D flags to the copy's Makefile; BPMF_FROM_GIT=<rev> first replaces the copy's sources with that revision's
(`head` = the committed kernels, for an A/B against the working tree)."""
import os, subprocess, sys
csrc = sys.argv[1]
root = os.path.dirname(os.path.dirname(csrc))
rev = os.environ.get("BPMF_FROM_GIT")
if rev:
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tar = subprocess.run(["git", "-C", repo, "archive", rev, "bpmf_amd/csrc", "include"], capture_output=True, check=True).stdout
    subprocess.run(["tar", "-x", "-C", root], input=tar, check=True)
defs = " ".join("-D" + d for d in os.environ.get("BPMF_DEFINES", "").split())
mk = os.path.join(csrc, "Makefile")
s = open(mk).read()
s = s.replace("CXXFLAGS := ", "CXXFLAGS := %s " % defs, 1)
open(mk, "w").write(s)
