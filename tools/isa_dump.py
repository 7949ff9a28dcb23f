#!/usr/bin/env python3
"""Development tool (no GPU needed): disassemble one kernel of the built library and print its instruction mix.
roc-obj does not run in this image (a Perl module is missing), so the gfx950 code object is carved out of the library's
offload bundle by hand.   python tools/isa_dump.py k_linesearch [--lib traj-opt-admm_amd/libtrajadmm.so] [--full]"""
import argparse, collections, os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("kernel", help="substring of the mangled kernel name, e.g. k_linesearch or k_xsolveILi43")
ap.add_argument("--lib", default=os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm.so"))
ap.add_argument("--full", action="store_true", help="print the disassembly itself, not only the mix")
a = ap.parse_args()
data = open(a.lib, "rb").read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
if i < 0: sys.exit("no offload bundle in " + a.lib)
n = struct.unpack_from("<Q", data, i + 24)[0]
off = i + 32; co = None
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
    t = data[off:off + tl].decode(); off += tl
    if "gfx950" in t: co = data[i + o:i + o + sz]
if co is None: sys.exit("no gfx950 code object in the bundle")
with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f: f.write(co); path = f.name
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", path], capture_output=True, text=True).stdout
os.unlink(path)
out, on = [], False
for line in dis.splitlines():
    if line.endswith(">:"):
        on = a.kernel in line
        if on: out.append(line)
        continue
    if on and line.strip(): out.append(line)
if not out: sys.exit("kernel not found")
if a.full: print("\n".join(out))
mix = collections.Counter(l.split()[0] for l in out if not l.endswith(">:"))
print(f"{sum(mix.values())} instructions")
for k, v in mix.most_common(25): print(f"{v:6d}  {k}")
