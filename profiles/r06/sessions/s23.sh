python tests/tools_dt_cu_spread.py 2>&1 | grep launch
