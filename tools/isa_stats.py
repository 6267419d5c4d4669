#!/usr/bin/env python
"""Register / scratch / LDS budget of every kernel in a hipcc object file or shared library (gfx950 code object notes):
   python tools/isa_stats.py bpmf_amd/csrc/k32.o [name-filter]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    out = []
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")                                # host ELF: the bundle sits in its .hip_fatbin section
    if subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], capture_output=True).returncode == 0 and os.path.exists(fat):
        path = fat
    targets = subprocess.run([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + path], capture_output=True, text=True).stdout.split()
    for t in targets:
        if "gfx" not in t:
            continue
        o = os.path.join(tmp, "co_%d.o" % len(out))
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + path, "--targets=" + t, "--output=" + o])
        out.append(o)
    return out


def main(path, flt=""):
    for co in code_objects(path):
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n  - (?=\.agpr_count:)", txt)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if flt and flt not in dem:
                continue
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            print("%-70s vgpr %3d agpr %3d sgpr %3d spill_v %3d scratch %4d lds %6d" % (dem[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
                  g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
