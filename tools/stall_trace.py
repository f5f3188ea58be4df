#!/usr/bin/env python3
"""Reads a rocprofv3 database of `bench.py --train` taken with --kernel-trace --hip-trace (no counters) and reports, for every gap
between two consecutive kernel dispatches that is longer than 20 ms, which HIP API calls of the host thread were in progress
during the gap (name, start and end relative to the gap).  VERDICT r02 #6: find what the host sits in during the ~100 ms-tick stalls.
  usage: python tools/stall_trace.py <results.db> [out.json]"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda stem: next(t for t in tabs if t.startswith(stem))
    strings = dict(db.execute("select id, string from '%s'" % T("rocpd_string")))
    kd = T("rocpd_kernel_dispatch")
    cols = [r[1] for r in db.execute("pragma table_info('%s')" % kd)]
    disp = db.execute("select start, end from '%s' order by start" % kd).fetchall()
    reg = T("rocpd_region")
    rcols = [r[1] for r in db.execute("pragma table_info('%s')" % reg)]
    regions = db.execute("select name_id, start, end from '%s' order by start" % reg).fetchall()
    gaps = []
    for (s0, e0), (s1, e1) in zip(disp, disp[1:]):
        if s1 - e0 > 20e6:
            gaps.append((e0, s1))
    t0 = disp[0][0]
    out = {"dispatches": len(disp), "api_calls": len(regions), "gaps_over_20ms": len(gaps), "gaps": []}
    for (a, b) in gaps[:12]:
        calls = []
        for nid, s, e in regions:
            if e < a or s > b:
                continue
            calls.append({"api": strings.get(nid, str(nid)), "start_ms_rel_gap": round((s - a) / 1e6, 3), "end_ms_rel_gap_end": round((e - b) / 1e6, 3),
                          "dur_ms": round((e - s) / 1e6, 3)})
        calls.sort(key=lambda c: -c["dur_ms"])
        out["gaps"].append({"gap_start_ms": round((a - t0) / 1e6, 3), "gap_ms": round((b - a) / 1e6, 3), "host_calls_in_progress": calls[:6]})
    js = json.dumps(out, indent=1)
    print(js)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(js)


if __name__ == "__main__":
    main()
