python tests/tools_dt_trace.py 640 480 1 2>&1 | tail -32
python tests/tools_dt_trace.py 640 480 9 2>&1 | tail -22
