for d in ${ABL:-0 2 4 8}; do echo "== MOE_DBG=$d"; MOE_DBG=$d timeout 120 python tools/gpu_diag.py layers 2>&1 | grep -E "B=12 layers|B=12 whole"; done
