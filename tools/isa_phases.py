#!/usr/bin/env python
"""Static per-phase instruction budget of ONE work item of k_sample1<32> (the headline kernel), from the ISA of the built object:

    python tools/isa_phases.py [bpmf_amd/csrc/k32.o]

The kernel is straight-line code around ONE loop (the 64-rating blocks of the Gram), so its phases can be cut at landmarks of
the instruction stream: the first MFMA (end of the prologue: index loads + the Philox / polar normal draw), the loop's back
edge, the DPP moves of assemble44 (the cross-block sums of the accumulators), the 2 x 16 v_rsq_f64 of the two-columns-per-step
Cholesky in finish_single, the v_rcp_f64 of the backward solve.  Printed per phase: VALU / MFMA / LDS / VMEM / SALU
instructions executed by a whole-column item with n ratings (loop body x blocks), and the VALU split by kind.
(The gate workgroup and the statistics riders of the fused launch are separate branches at the top of the kernel: not counted.)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(obj, symbol_part):
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    co = os.path.join(tmp, "k.co")
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    out, on = [], False
    for l in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            on = symbol_part in m.group(1)
            continue
        if on:
            q = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
            if q:
                out.append((q.group(1), q.group(2), int(q.group(3), 16)))
    return out


def cls(m):
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("v_"): return "valu"
    if m.startswith("ds_"): return "lds"
    if m.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if m.startswith("s_"): return "salu"
    return "other"


def kind(m):
    if "dpp" in m: return "dpp move"
    if m.startswith("v_readlane") or m.startswith("v_readfirstlane"): return "readlane"
    if m.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")): return "f64 arithmetic"
    if m.startswith(("v_rsq", "v_rcp", "v_sqrt", "v_log", "v_exp", "v_frexp", "v_ldexp", "v_cvt", "v_trunc", "v_fract", "v_floor", "v_rndne")): return "transcendental / convert"
    if m.startswith(("v_mul_hi", "v_mul_lo", "v_mad_u64", "v_mad_u32", "v_mad_i32", "v_mul_u32", "v_mul_i32")): return "integer multiply"
    if m.startswith(("v_cndmask", "v_cmp")): return "select / compare"
    if m.startswith(("v_mov", "v_accvgpr")): return "move"
    return "32-bit logic / add / address"


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "bpmf_amd", "csrc", "k32.o")
    ins = disassemble(obj, "k_sample1ILi32")
    n = len(ins)
    addr = {a: i for i, (m, o, a) in enumerate(ins)}
    mf = [i for i, (m, o, a) in enumerate(ins) if m.startswith("v_mfma")]
    rsq = [i for i, (m, o, a) in enumerate(ins) if m.startswith("v_rsq_f64")]
    dpp = [i for i, (m, o, a) in enumerate(ins) if "dpp" in m]
    # the Gram loop: the backward branch whose span holds the most MFMAs
    loops = []
    for i, (m, o, a) in enumerate(ins):
        if m.startswith("s_cbranch") or m == "s_branch":
            t = re.search(r"<.*\+0x([0-9a-f]+)>", o)
            if not t:
                continue
            # objdump prints the target as symbol+offset: the function starts at ins[0]'s address
            tgt = ins[0][2] + int(t.group(1), 16) - (ins[0][2] - ins[0][2])
            tgt_i = None
            base = None
    # simpler: the loop body = from the first MFMA-dense run's start to the back edge `s_cbranch_scc1` that follows >= 100 MFMAs
    back = [i for i, (m, o, a) in enumerate(ins) if m in ("s_cbranch_scc1", "s_cbranch_scc0", "s_cbranch_vccnz", "s_cbranch_vccz") and i > mf[0]]
    body_end = None
    for b in back:
        if sum(1 for q in mf if q < b) >= 140:
            body_end = b
            break
    # loop start: the instruction the branch jumps back to -- approximate as the first MFMA minus its operand prologue
    # (find the label by the branch's signed offset, simm16 dwords from the next instruction)
    word = ins[body_end][1]
    m16 = re.match(r"(\d+)", word)
    off = int(m16.group(1))
    if off >= 32768:
        off -= 65536
    tgt_addr = ins[body_end + 1][2] + 4 * off
    body_start = addr[tgt_addr]
    draw_rsq = [i for i in rsq if i < body_start]
    fin_rsq = [i for i in rsq if i > body_end]
    # pairs of v_rsq of the factorisation: 32 of them, the first after the DPP block of assemble44
    asm_start = min(i for i in dpp if i > body_end)
    fac_rsq = [i for i in fin_rsq if i > asm_start]
    fac_start = fac_rsq[0] - 12                                   # (the pivot readlanes ahead of the first 1/sqrt)
    rcp = [i for i, (m, o, a) in enumerate(ins) if m.startswith("v_rcp_f64") and i > fac_rsq[-1]]
    bwd_start = (rcp[0] - 4) if rcp else fac_rsq[-1] + 60
    # landmarks of the rest: the partial-sum loads of a chunked column's last arriver and the SECOND normal draw (the one a
    # last arriver runs: its column has no whole-column item) sit between the Gram and the assembly; the statistics-rider and
    # gate bodies of the fused launch are the code before the item prologue and after the final store
    logs = [i for i, (m, o, a) in enumerate(ins) if m.startswith(("v_log", "v_frexp_mant"))]
    draw0 = min(logs) - 110                                           # Philox rounds ahead of the first logarithm
    draw0 = max(i for i, (m, o, a) in enumerate(ins) if i < draw0 and (m.startswith("s_cbranch") or m == "s_branch")) + 1
    park0 = min(i for i, (m, o, a) in enumerate(ins) if i > body_end and m.startswith("global_atomic") or (i > body_end and m.startswith("global_store")))
    stores = [i for i, (m, o, a) in enumerate(ins) if m.startswith("global_store") and i > bwd_start]
    end_item = stores[0] + 12
    cuts = [("item words, two index blocks, normal draw (Philox rounds, polar test, polar_mult) of a whole column", draw0, body_start, 1),
            ("Gram: one 64-rating block of the loop (gathers one group ahead, 4 x 36 MFMAs, rhs FMAs)", body_start, body_end + 1, None),
            ("Gram: last block of the chunk (1-4 groups of 16 ratings)", body_end + 1, park0 - 30, 1),
            ("[chunk hand-over: park partials, ticket, sum partials, second normal draw -- NOT executed by a whole column]", park0 - 30, asm_start, 0),
            ("assembly: cross-block sums (DPP), G -> LDS (mirrored), Lambda* = LambdaF + alpha G, rhs", asm_start, fac_start, 1),
            ("factorisation: 16 steps of two columns (10 readlanes, 2 x 1/sqrt + Halley, scale + publish, rank-2 update, forward solve)", fac_start, bwd_start, 1),
            ("backward solve (31 readlane broadcasts), store, failure check", bwd_start, end_item, 1)]
    nr = int(os.environ.get("RATINGS", "160"))
    blocks = max(0, (nr - 1) // 64)                               # full loop iterations; the last block runs in the tail code
    print("k_sample1<32>: %d instructions in the kernel; budget of a whole-column item with %d ratings (%d loop blocks + tail)" % (n, nr, blocks))
    print("%-118s %6s %6s %6s %6s %6s" % ("phase (static count x times executed)", "VALU", "MFMA", "LDS", "VMEM", "SALU"))
    tot = collections.Counter()
    for name, a0, a1, times in cuts:
        t = blocks if times is None else times
        c = collections.Counter(cls(m) for m, o, a in ins[a0:a1])
        if times == 0:
            print("%-118s   (static %d: %s)" % (name[:118], a1 - a0, ", ".join("%s %d" % kv for kv in sorted(c.items()))))
            continue
        print("%-118s %6d %6d %6d %6d %6d   (static %d, x%d)" % (name[:118], c["valu"] * t, c["mfma"] * t, c["lds"] * t, c["vmem"] * t, c["salu"] * t, a1 - a0, t))
        for k in c:
            tot[k] += c[k] * t
        kk = collections.Counter(kind(m) for m, o, a in ins[a0:a1] if cls(m) == "valu")
        print("      VALU by kind: " + ", ".join("%s %d" % (k, v * t) for k, v in kk.most_common()))
    print("%-118s %6d %6d %6d %6d %6d" % ("total", tot["valu"], tot["mfma"], tot["lds"], tot["vmem"], tot["salu"]))
    print("issue estimate (measured rates, DESIGN.md section 4: f64 VALU ~5 cycles, 32-bit ~3.3, v_mfma_f64_4x4x4 ~18, MFMA and VALU issue ADD on a SIMD):")
    print("      VALU %d x ~4.5 = %d cycles, MFMA %d x 18 = %d cycles per item; 5 038 items on 1 024 SIMDs" % (tot["valu"], tot["valu"] * 4.5, tot["mfma"], tot["mfma"] * 18))


if __name__ == "__main__":
    main()
