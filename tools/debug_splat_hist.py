"""Rectangle-size classes of the photons of the bench workloads (stats build only)."""
import os, sys, json
os.environ["EVPLP_LIB"] = os.path.join(os.path.dirname(__file__), "..", "evplp_amd", "lib", "libevplp_hip_stats.so")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import subprocess
for wl in ("evplp", "ppm"):
    out = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=dict(os.environ, EVPLP_DUMP_SPLAT_HIST="1"))
    print(wl, out.stderr[-2000:] if out.returncode else "", [l for l in out.stdout.splitlines() if l.startswith("HIST")])
