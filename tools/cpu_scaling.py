"""Thread scaling of the CPU oracle's HSQ compress on this box's host: `bench.py --cpu-scaling` (the CPU-baseline leg of
bench.py is the one place outside tests/ that runs the oracle; this is a wrapper around it)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-scaling"]))
