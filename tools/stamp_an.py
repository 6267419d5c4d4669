"""Reads the STAMP lines of the trace variant (tools/stamp_patch.py) from stdin: per sampler launch the
first / last workgroup start, the last end, and the end of the gate kernel that released it."""
import sys, re
rows = []
for l in sys.stdin:
    m = re.match(r"STAMP slot\s+(\d+) side (\S+) first_start (\d+) last_start (\d+) last_end (\d+)(?: gate_end (\d+))?", l)
    if m: rows.append((m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6) or 0)))
rows.sort(key=lambda r: r[1])
prev = None
for side, fs, ls, le, ge in rows:
    print("side %s: span %.1f us, last workgroup started at +%.1f us; previous launch's last end -> first start: %s us; its gate kernel ended %s us before that end" % (
        side[-5:], (le - fs) / 100.0, (ls - fs) / 100.0, "%.1f" % ((fs - prev) / 100.0) if prev else "-",
        "%.1f" % ((prev - ge) / 100.0) if (prev and ge) else "-"))
    prev = le
