# same-box A/B: alternate variants, 2 rounds
for r in 1 2; do
  for v in 1 2; do echo "ATTN NST=$v"; SVG_ATTN_NST=$v python tools/kbench.py attn --b 16 2>&1 | grep -E "^ *4096 +4096|^ *1024 +1024"; done
  for v in 0 4; do echo "GEMM DBG=$v"; SVG_GEMM_DBG=$v python tools/kbench.py gemm --b 16 2>&1 | grep -E "^ *65536 +320 +320|^ *65536 +320 +1280|^ *65536 +2560|^ *16384 +640 +640 |^ *4096 +1280 +1280"; done
done
for v in 0 4; do echo "BENCH DBG=$v"; SVG_GEMM_DBG=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for v in 1 2; do echo "BENCH NST=$v"; SVG_ATTN_NST=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
