for d in 0 1 2 3; do echo "DBG=$d"; SVG_GEMM_DBG=$d python tools/kbench.py gemm --b 16 2>&1 | grep -E "^ *65536 +320 +320|^ *65536 +320 +1280|^ *65536 +2560|^ *16384 +640 +640 "; done
