#!/usr/bin/env python3
"""Run a script of the reference checkout (its main.py) on the MI355X implementation, unedited:

    cd /path/to/gradient-quantization
    python /root/repo/gradient-quantization_amd/run_reference.py main.py --quantizer hsq --network resnet50 ...

`python main.py` puts the script's own directory first on sys.path, so the reference's
`compressors/` and `quantizers/` packages would win over anything on PYTHONPATH.  This launcher puts
THIS directory (with the shadowing `compressors/`, `quantizers/` packages) first, the script's
directory second (for `models`, `dataloaders`, `logger`, ...), and runs the script as __main__.
"""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        sys.exit("usage: run_reference.py <script.py> [script args...]")
    here = os.path.dirname(os.path.abspath(__file__))
    script = os.path.abspath(sys.argv[1])
    sys.path[:] = [here, os.path.dirname(script)] + [p for p in sys.path if p not in ("", here, os.path.dirname(script))]
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
