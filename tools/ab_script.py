"""A/B of any timing script of tools/ for several library builds on ONE box, alternated twice:
    python tools/ab_script.py tools/hsq_batched_r.py product tools/exp/libgq_X.so ...   ('product' = the in-tree library)
prints the script's lines that hold ' us'."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
script, libs = sys.argv[1], sys.argv[2:] or ["product"]
for rep in range(2):
    for l in libs:
        env = dict(os.environ)
        if l != "product":
            env["GQ_LIB_PATH"] = os.path.join(ROOT, l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, script)], env=env, capture_output=True, text=True).stdout
        for ln in out.splitlines():
            if " us" in ln:
                print("%-28s %s" % (l, ln[:110]), flush=True)
