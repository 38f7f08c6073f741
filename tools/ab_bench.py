import os, sys, json, io, contextlib, runpy
sys.path.insert(0, ".")
from score_based_channels_amd import _lib
if os.environ.get("SBC_LIB_OVERRIDE"): _lib.LIB_PATH = os.environ["SBC_LIB_OVERRIDE"]
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "30"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(os.environ.get("SBC_LIB_OVERRIDE", "new"), round(d["value"], 2), round(d["ms_per_step"], 3))
