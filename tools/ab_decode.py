"""A/B of the decode-mean (tools/decode_r.py) for several library builds on ONE box:
    python tools/ab_decode.py product tools/exp/libgq_X.so ...   ('product' = the in-tree library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["product"]
for rep in range(2):
    for l in libs:
        env = dict(os.environ)
        if l != "product":
            env["GQ_LIB_PATH"] = os.path.join(ROOT, l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "decode_r.py"), "1", "4", "8", "16"], env=env,
                             capture_output=True, text=True).stdout
        times = [ln.split(":")[1].split("us")[0].strip() for ln in out.splitlines() if ln.startswith("R=")]
        print("%-36s R=1/4/8/16: %s us" % (l, " / ".join(times)), flush=True)
