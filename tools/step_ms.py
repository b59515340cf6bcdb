"""prints ms/step of the default bench.py workload (stdin-free helper for tools/variants.sh)"""
import json, subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", "30", "--warmup", "8"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1])
print(f"{d['ms_per_step']:.3f} ms/step  {d['value']:.1f} {d['unit']}  roofline {d['roofline']['frac']:.3f}")
