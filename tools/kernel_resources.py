#!/usr/bin/env python3
"""VGPRs / scratch (spills) / occupancy of every kernel of libqecmc, from the resource remarks the build keeps in
csrc/build/<unit>.res (`make` writes them; hipcc cross-compiles on the CPU).

    tools/kernel_resources.py [unit ...]        e.g.  tools/kernel_resources.py ladder_biased
"""
import glob
import os
import re
import sys

BUILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mcmc-qec-toric-rl_amd", "csrc", "build")
FLAGS = ["conv", "gsplit", "biased", "scan", "gentop", "uset", "alpha", "pre", "delut", "queue", "ssw"]   # LadderFlag bit order
CODES = ["toric", "xzzx", "rotated", "planar"]


def kernel_label(mangled):
    """ladder_kernel<MAXT, MINW, CODE, FLAGS> -> 'ladder<512,8,toric: gsplit|delut|ssw>'; other kernels by name."""
    w = re.search(r"ladder_wu_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELi(\d+)ELb([01])E", mangled)
    if w:   # ladder_wu_kernel<MAXT, MINW, CODE, WV, CONV, QUEUE, IT, ALPHA> -> 'wave<512,8,toric: 12 words, conv, queue, iters 10>'
        maxt, minw, code, wv, conv, queue, it, alpha = (int(x) for x in w.groups())
        return "wave<%d,%d,%s: %d words%s%s%s%s>" % (maxt, minw, CODES[code], wv, ", conv" if conv else "", ", queue" if queue else "", ", alpha" if alpha else "",
                                                   ", iters %d" % it if it else "")
    m = re.search(r"ladder_kernelILi(\d+)ELi(\d+)ELi(\d+)ELj(\d+)E", mangled)
    if not m:
        m2 = re.match(r"_ZN5qecmc\d+([A-Za-z_0-9]+?)(?:I|E)", mangled)
        return m2.group(1) if m2 else mangled[:60]
    maxt, minw, code, fl = (int(x) for x in m.groups())
    names = [n for i, n in enumerate(FLAGS) if fl >> i & 1]
    return "ladder<%d,%d,%s: %s>" % (maxt, minw, CODES[code], "|".join(names) or "plain")


def parse(path):
    """[{unit, kernel, label, VGPRs, AGPRs, SGPRs, ScratchSize, Occupancy}, ...] of one .res file"""
    rows, cur = [], None
    keys = {"VGPRs": r"\sVGPRs: (\d+)", "AGPRs": r"\sAGPRs: (\d+)", "SGPRs": r"\sSGPRs: (\d+)",
            "ScratchSize": r"ScratchSize \[bytes/lane\]: (\d+)", "Occupancy": r"Occupancy \[waves/SIMD\]: (\d+)"}
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"unit": os.path.basename(path)[:-4], "kernel": m.group(1), "label": kernel_label(m.group(1))}
            rows.append(cur)
            continue
        if cur is not None:
            for k, pat in keys.items():
                mm = re.search(pat, line)
                if mm:
                    cur[k] = int(mm.group(1))
    return [r for r in rows if "ScratchSize" in r]


def all_rows(units=None):
    files = sorted(glob.glob(os.path.join(BUILD, "*.res")))
    if units:
        files = [f for f in files if os.path.basename(f)[:-4] in units]
    return [r for f in files for r in parse(f)]


if __name__ == "__main__":
    rows = all_rows([u.replace(".hip", "") for u in sys.argv[1:]])
    if not rows:
        sys.exit("no resource remarks under %s: run `make -C mcmc-qec-toric-rl_amd/csrc` first" % BUILD)
    for r in rows:
        print("%-16s %-62s VGPRs %3d  scratch %4d B/lane  occupancy %d" % (r["unit"], r["label"], r["VGPRs"], r["ScratchSize"], r["Occupancy"]))
