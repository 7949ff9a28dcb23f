#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a built library, read from the gfx950 code object's metadata.
Usage: python tools/kernel_resources.py [path/to/lib.so] [regex]      (development aid; needs no GPU)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def resources(lib):
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        with open(lib, "rb") as f, open(so, "wb") as g:
            g.write(f.read())
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True, capture_output=True)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f]
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, cos[0])], check=True, capture_output=True, text=True).stdout
    name, d = None, {}
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1); d[name] = {}
        for k in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count"):
            m = re.search(r"\." + k + r":\s+(\d+)", line)
            if m and name:
                d[name][k] = int(m.group(1))
    return d


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else "."
    print("%-64s %5s %5s %5s %8s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "spill"))
    for n, v in sorted(resources(lib).items()):
        if re.search(pat, n):
            print("%-64s %5d %5d %5d %8d %8d %6d" % (n[:64], v.get("vgpr_count", 0), v.get("agpr_count", 0), v.get("sgpr_count", 0),
                                                     v.get("private_segment_fixed_size", 0), v.get("group_segment_fixed_size", 0), v.get("vgpr_spill_count", 0)))
