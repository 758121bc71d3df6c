import os, sys, subprocess
for t in ["2x28","1x28","1x56","2x14","4x14","1x14","2x56","4x28"]:
    env = dict(os.environ, SHL_MI355X_STEMDW_TILE=t)
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--detail", "--steps", "100"], capture_output=True, text=True, env=env)
    line = [l for l in r.stderr.splitlines() if "stemdw" in l]
    print(t, line[0].split()[-6:] if line else r.stderr[-300:])
