#!/usr/bin/env python3
"""profiles/<tag>_parity.md from the artefacts a `pytest -m gpu` run leaves under gpurun_out/ (parity_margins.json: every
network-level error the tests measured, with its tolerance; ddim_drift.json: the per-step tables of the 50-step DDIM loop).
usage: python tools/make_parity_report.py r02"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
m = json.load(open(os.path.join(ROOT, "gpurun_out", "parity_margins.json")))
out = ["# Parity margins measured on MI355X (%s)" % tag, "",
       "Every network-level comparison of `pytest -m gpu` (HIP path through the C ABI vs the CPU oracle on identical seeded weights,",
       "inputs and noise), as reported by `tests/conftest.py:margin`.  Tolerances in the tests are <= 3x the values below, except the",
       "free-running many-step DDIM comparisons, which are bounded by the saturation level of a chaotic map (see the table further down).", "",
       "| check | measured | unit | tolerance | margin |", "|---|---|---|---|---|"]
seen = {}
for e in m:
    k = e["name"]
    if k in seen and seen[k]["measured"] >= e["measured"]:
        continue
    seen[k] = e
for k, e in seen.items():
    out.append("| %s | %.3e | %s | %.1e | x%.1f |" % (k, e["measured"], e["unit"], e["tolerance"], e["tolerance"] / max(e["measured"], 1e-30)))
p = os.path.join(ROOT, "gpurun_out", "ddim_drift.json")
if os.path.exists(p):
    d = json.load(open(p))
    out += ["", "## The 50-step DDIM loop of configs[2], full-size UNet, step by step (tests/test_configs_gpu.py)", "",
            "Teacher-forced: every step starts from the ORACLE's latent z_k (fixture tests/golden/sd_cfg2_frame.pt) and is compared with the",
            "oracle's z_{k+1} — the per-step arithmetic error of UNet call + scheduler step, no accumulation:", "",
            "| step k | " + " | ".join(str(k) for k in range(0, 50, 5)) + " | 49 | worst |",
            "|---|" + "---|" * 12]
    tf = d["teacher_forced_rel_l2_per_step"]
    out.append("| rel-L2 of z_{k+1} | " + " | ".join("%.1e" % tf[k] for k in range(0, 50, 5)) + " | %.1e | %.1e |" % (tf[49], max(tf)))
    out += ["", "Free-running from the oracle's starting latent (HIP vs oracle) next to the growth of a 1e-3 (rel-L2) perturbation of that",
            "latent through the SAME HIP loop (HIP vs HIP): the two curves coincide — the distance after many steps measures the conditioning",
            "of this seeded random-weight network (a chaotic map in its first ten steps, where x0 is divided by sqrt(alpha_t) <= 0.2), not the",
            "arithmetic:", "", "| after k steps | " + " | ".join(str(k) for k in d["steps"]) + " |", "|---|" + "---|" * len(d["steps"]),
            "| HIP vs fp32 oracle | " + " | ".join("%.2e" % v for v in d["free_running_rel_l2_vs_oracle"]) + " |",
            "| 1e-3 perturbation, HIP vs HIP | " + " | ".join("%.2e" % v for v in d["growth_of_1e-3_perturbation_hip_vs_hip"]) + " |"]
open(os.path.join(ROOT, "profiles", "%s_parity.md" % tag), "w").write("\n".join(out) + "\n")
print("wrote profiles/%s_parity.md (%d checks)" % (tag, len(seen)))
