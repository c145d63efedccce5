"""Does WHERE in device memory a process's buffers lie decide its placement class?  (round 5, profiles/r05/README.md)
For each ballast size (GB, argv): a fresh process first takes that much device memory straight from the HIP runtime and
keeps it, then runs the plain bench -- so the library's buffers come from another region of the device than without.
    python tools/ballast_check.py 0 60 120 180 0"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, runpy, sys
gb = int(sys.argv[1])
if gb:
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30))
    assert rc == 0, rc
    hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30))          # (touched: the pages are really taken)
    hip.hipDeviceSynchronize()
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "20", "--warmup", "3"]
runpy.run_path(sys.argv[0], run_name="__main__")
'''
for gb in [int(x) for x in sys.argv[1:]] or [0, 60, 120, 180, 0]:
    out = subprocess.run([sys.executable, "-c", CHILD, str(gb)], cwd=ROOT, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        st = d["roofline"]["stage_ms_per_step_calibration"]
        print(f"ballast {gb:4d} GB: {d['value'] / 1e9:6.2f} G k-mers/s  {d['ms_per_step']:.3f} ms/step  scatter1 {st['scatter1']:.3f}  "
              f"scatter2 {st['scatter2']:.3f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"ballast {gb} GB: no line ({e}); stderr tail: {out.stderr[-300:]}", flush=True)
