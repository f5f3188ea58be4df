"""Summarise the per-stage stamps of a -DWK_STAMP build (SVG_XF_WALK_STAMPS=<file>): where a stage's time goes.
usage: python tools/xf_walk_stamps.py <file> [launch_us]
s_memtime counts shader clocks on this part (not the 100 MHz reference clock): without `launch_us` (the launch's duration from rocprofv3 or
events) the table is in units of 100 ticks — divide by ~22 for microseconds at 2.2 GHz; with it, everything is scaled to microseconds."""
import sys
from collections import defaultdict

KIND = {0: "gemm", 1: "reduce", 2: "reduce+LN", 3: "attention", 4: "embed"}
rows = []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    a, b = line.split("|")
    i, kind, bar, M, N, K = [int(x) for x in a.split()]
    t = [int(x) for x in b.split()]
    rows.append((kind, bar, M, N, K, t))
acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for kind, bar, M, N, K, t in rows:
    key = (KIND[kind], "bar" if bar else "nobar", N, K) if kind == 0 else (KIND[kind], "bar" if bar else "nobar", 0, 0)
    e = acc[key]
    e[0] += 1
    if bar:
        e[1] += (t[1] - t[0]) * 0.01
        e[2] += (t[2] - t[1]) * 0.01
    e[3] += (t[3] - t[2]) * 0.01
total = (rows[-1][5][3] - rows[0][5][0]) * 0.01
scale = float(sys.argv[2]) / total if len(sys.argv) > 2 else 1.0
unit = "us" if len(sys.argv) > 2 else "x100 ticks"
print("whole launch (workgroup 0): %.1f %s over %d stages" % (total * scale, unit, len(rows)))
print("%-34s %5s %10s %10s %10s %10s" % ("stage", "n", "drain us", "barrier us", "work us", "sum us"))
for key, e in sorted(acc.items(), key=lambda kv: -(kv[1][1] + kv[1][2] + kv[1][3])):
    n = e[0]
    print("%-34s %5d %10.2f %10.2f %10.2f %10.1f" % ("%s %s %s" % (key[0], key[1], ("N%d K%d" % (key[2], key[3])) if key[2] else ""), n, scale * e[1] / n, scale * e[2] / n,
                                                    scale * e[3] / n, scale * (e[1] + e[2] + e[3])))
