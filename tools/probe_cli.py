"""Runs the CLI on a synthetic three-cloud case with the reference's own 11^6 grid (no <search> grid nodes),
once per search mode, and prints wall times and the first lines of results.txt."""
import os, sys, time, subprocess, tempfile, pathlib, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cameracalibration_amd as cc
from tests.test_gpu_cli import build_case, CLI
tmp = pathlib.Path(tempfile.mkdtemp())
cfg = build_case(tmp, n_points=int(sys.argv[1]) if len(sys.argv) > 1 else 200000)["cfg"]
text = open(cfg).read()
text = text.replace("<gd_max_iterations>12</gd_max_iterations>", "")
text = re.sub(r"<translation_mm>.*?</translation_mm>", "", text, flags=re.S)
text = re.sub(r"<rotation_deg>.*?</rotation_deg>", "", text, flags=re.S)
for mode in ("dense", "separable", "dense", "separable"):
    t = text.replace("<search>", f"<search><mode>{mode}</mode>")
    path = tmp / f"config_{mode}.xml"
    path.write_text(t)
    wd = tmp / mode
    os.makedirs(wd, exist_ok=True)
    t0 = time.time()
    r = subprocess.run([CLI, str(path)], cwd=wd, capture_output=True, text=True)
    print(mode, "rc", r.returncode, f"{time.time()-t0:.2f} s", flush=True)
    print("".join(open(wd / "results.txt").readlines()[1:8]), flush=True)
    if r.returncode: print(r.stderr[-500:])
