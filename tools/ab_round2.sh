# same-box A/B of the round-2 changes to the UNet step: default build vs the switches that restore the round-1 paths
# (GroupNorm statistics pass + materialised concat + separate ff1 / ff2).  usage (GPU box): bash tools/ab_round2.sh
for rep in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export SVG_GN_EPI=0 SVG_GN_FOLD=0 SVG_FF_FUSED=0; else unset SVG_GN_EPI SVG_GN_FOLD SVG_FF_FUSED; fi
    timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${v}_$rep.log 2>&1 || exit 1
    python3 - <<PY
import json
for l in open("gpurun_out/ab_${v}_$rep.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("$v $rep: %.2f frames/s  unet_step %.2f ms  " % (d["value"], r["unet_step"]["ms_per_call"]) + "  ".join("%s %.0f" % (k, v["ms"]) for k, v in r["by_family"].items()))
PY
  done
done
