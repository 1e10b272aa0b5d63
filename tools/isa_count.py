"""Instruction census of the prefilter encode's tile loop, straight from the compiler's ISA.

    python tools/isa_count.py [--kernel MANGLED_SUBSTR] [--file csrc/hsq_encode_pf.hip] [--list]

Compiles the file with build.py's flags + --save-temps into a scratch directory, cuts the kernel's steady-state tile
loop out of the gfx950 assembly (the outermost loop body, minus the rare wave-wide fix-up loop nested in it and the
paths guarded by s_cbranch_exec* that the common tile does not take are kept -- they are skipped at run time but cost
their scalar branch only), and counts instructions by issue class.  Slot costs are the measured ones of
profiles/r02_b_valu_op_rates.txt (two waves per SIMD): "half" = v_fma/fmac/mul/add/sub_f32, v_and_b32, v_mov_b32,
v_add_u32 with all-VGPR operands (1.25 ns), "full" = every other VALU op (1.95 ns), swap = v_permlane32_swap (3.5 ns).
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gradient-quantization_amd")
HALF = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_mov_b32",
        "v_add_u32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32"}
NS = {"half": 1.25, "full": 1.95, "swap": 3.5, "mfma": 8 / 2.1}     # ns of issue per wave-instruction at ~2.1 GHz


def build_asm(src, extra):
    sys.path.insert(0, PKG)
    import importlib.util
    spec = importlib.util.spec_from_file_location("gq_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    d = tempfile.mkdtemp(prefix="gq_isa_")
    flags = [f for f in b.FLAGS if f != "-shared"] + b.EXTRA.get(os.path.basename(src), []) + extra
    subprocess.check_call([b.hipcc()] + flags + ["--save-temps", "-c", src, "-o", os.path.join(d, "x.o")], cwd=d,
                          stderr=subprocess.DEVNULL)
    s = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    return open(os.path.join(d, s)).read()


def kernel_body(txt, match):
    names = [m.group(1) for m in re.finditer(r"^(_Z\w+):\s*(;.*)?$", txt, re.M)]
    names = [n for n in names if match in n]
    if not names:
        raise SystemExit("no kernel matches %r" % match)
    name = names[0]
    i = txt.index("\n" + name + ":")
    j = txt.index(".Lfunc_end", i)
    meta = {}
    k = txt.index(".amdhsa_kernel " + name)
    for key in ("next_free_vgpr", "accum_offset"):
        m = re.search(r"\.amdhsa_%s (\d+)" % key, txt[k:k + 4000])
        meta[key] = int(m.group(1)) if m else None
    m = re.search(r"; ScratchSize: (\d+)", txt[j:j + 3000])
    meta["scratch"] = int(m.group(1)) if m else None
    m = re.search(r"; NumVgprs: (\d+)", txt[j:j + 3000])
    meta["vgprs"] = int(m.group(1)) if m else None
    return name, txt[i:j].split("\n"), meta


def loops(lines):
    """(start, end, depth) of every natural loop: a backward branch to a label."""
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    out = []
    for n, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) <= n:
            out.append((labels[m.group(1)], n))
    return sorted(set(out))


def classify(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_permlane"):
        return "swap"
    if op.startswith("v_"):
        op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if op in HALF and not re.search(r"\bs\d+|\bs\[|0x|\blit\(|\bvcc|, -?\d+(\.\d+)?($|,)", line.split(None, 1)[1] if " " in line.strip() else ""):
            return "half"
        return "full"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", default=os.path.join(PKG, "csrc", "hsq_encode_pf.hip"))
    ap.add_argument("--kernel", default="hsq_encode_pf_kernelIhLi16ELb0ELb0ELb1ELi16ELi8EE")
    ap.add_argument("--list", action="store_true", help="print the per-mnemonic table too")
    ap.add_argument("--extra", default="", help="extra compiler flags, e.g. -DPF_GROUP8=1")
    ap.add_argument("--asm", default=None, help="an existing .s instead of compiling")
    ap.add_argument("--json", default=None, help="also write the counts as JSON (profiles/r04_encode_isa_floor.json: bench.py's issue_floor_ms)")
    a = ap.parse_args()
    txt = open(a.asm).read() if a.asm else build_asm(a.file, a.extra.split())
    name, lines, meta = kernel_body(txt, a.kernel)
    lp = loops(lines)
    outer = max(lp, key=lambda se: se[1] - se[0])                  # the tile loop
    inner = [se for se in lp if se != outer and outer[0] <= se[0] and se[1] <= outer[1]]
    skip = set()
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    for s, e in inner:                                             # the rare fix-up loop ...
        skip.update(range(s, e + 1))
        for n in range(s - 1, outer[0], -1):                       # ... and its setup block: from the forward branch that
            m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", lines[n])   # skips the loop when no lane is flagged
            if m and labels.get(m.group(1), -1) > e:
                skip.update(range(n + 1, s))
                break
            if re.match(r"^\.LBB", lines[n]) and n < s - 60:
                break
    cls = collections.Counter()
    per = collections.Counter()
    for n in range(outer[0], outer[1] + 1):
        if n in skip:
            continue
        l = lines[n].split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        c = classify(op, l)
        cls[c] += 1
        per[(c, re.sub(r"_(e32|e64)$", "", op))] += 1
    print("kernel %s" % name)
    print("registers: %s" % meta)
    print("tile loop: asm lines %d..%d, nested loops excluded: %s" % (outer[0], outer[1], inner))
    valu = cls["half"] + cls["full"] + cls["swap"]
    ns = sum(cls[k] * NS[k] for k in NS)
    print("VALU %d (half-slot %d, full-slot %d, permlane swaps %d), MFMA %d, LDS %d, VMEM %d, SALU %d"
          % (valu, cls["half"], cls["full"], cls["swap"], cls["mfma"], cls["lds"], cls["vmem"], cls["salu"]))
    print("issue time per tile and wave at two waves per SIMD: %.0f ns VALU + %.0f ns MFMA issue = %.0f ns"
          % (ns - cls["mfma"] * NS["mfma"], cls["mfma"] * NS["mfma"], ns))
    if a.json:
        import json
        json.dump({"kernel": name, "vgprs": meta["vgprs"], "scratch": meta["scratch"], "valu": valu, "valu_half_slot": cls["half"],
                   "valu_full_slot": cls["full"], "permlane_swaps": cls["swap"], "mfma": cls["mfma"], "lds": cls["lds"],
                   "vmem": cls["vmem"], "salu": cls["salu"], "subvectors_per_tile": 64,
                   "issue_cycles_per_tile": 4 * valu + 8 * cls["mfma"],
                   "note": "instructions of the steady-state tile loop per wave and 64-subvector tile, from the compiler's ISA "
                           "(tools/isa_count.py; the second pass and the exact scans, nested loops, are not in it; ~40 of the VALU "
                           "instructions sit in branches most tiles skip); issue_cycles_per_tile = 4 cycles per VALU "
                           "instruction + 8 per MFMA (MI355X_MICROARCH.md, vector-instruction issue cost): a SIMD issues for its "
                           "two waves in turn, so the launch cannot take less than tiles x issue_cycles / SIMDs / clock"},
                  open(a.json, "w"), indent=1)
    if a.list:
        for (c, op), k in sorted(per.items(), key=lambda x: (-x[1], x[0])):
            if c in ("half", "full", "swap", "mfma", "lds", "vmem"):
                print("  %-5s %-28s %d" % (c, op, k))


if __name__ == "__main__":
    main()
