#!/usr/bin/env python3
"""Development tool (no GPU needed): the instruction sequences of the signalling sites of the shipped code object -- what DESIGN.md section 3 ("Synchronisation
protocols") quotes and tests/test_abi_and_host.py::test_signalling_sites_keep_their_order checks.  For every kernel of the iteration chain: each producer site
(s_nop 0x2a1 ... 0x2a2, with the stores in front of it) and the head of each wait (s_nop 0x2b1 up to the poll and its compare).
  python tools/isa_protocols.py [regex of kernels] [--lib traj-opt-admm_amd/libtrajadmm.so] > profiles/roundN_signalling_isa.txt"""
import argparse, os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("pattern", nargs="?", default=r"tj\d+(k_grad|k_xsolveILi43|k_linesearch|k_frontILi1ELb1|k_midILi1ELb1|k_ccd_leanILi1|k_keepE|k_xs_gate|k_fa_gate)")
ap.add_argument("--lib", default=os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm.so"))
ap.add_argument("--before", type=int, default=6, help="instructions shown in front of a producer marker (the record stores)")
a = ap.parse_args()
data = open(a.lib, "rb").read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", data, i + 24)[0]
off = i + 32; co = None
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
    t = data[off:off + tl].decode(); off += tl
    if "gfx950" in t: co = data[i + o:i + o + sz]
with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f: f.write(co); path = f.name
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", path], capture_output=True, text=True).stdout
os.unlink(path)
body, cur = {}, None
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m: cur = m.group(1); body[cur] = []
    elif cur and line.strip(): body[cur].append(line.split("//")[0].rstrip().strip())
mem = ("global_", "flat_", "buffer_", "scratch_")
for k in sorted(body):
    if not re.search(a.pattern, k): continue
    ins = body[k]
    print(f"== {k}")
    for j, t in enumerate(ins):
        if t.startswith("s_nop 0x2a1"):
            e = j
            while e < len(ins) and not ins[e].startswith("s_nop 0x2a2") and not ins[e].startswith("s_endpgm"): e += 1
            lo = j
            cnt = 0
            while lo > 0 and cnt < a.before:   # the last memory writes in front of the marker
                lo -= 1
                if ins[lo].startswith(mem): cnt += 1
            print("  -- producer site (stores in front of the marker, then the marked region; non-memory instructions elided)")
            for t2 in ins[lo:j]:
                if t2.startswith(mem): print("       " + t2)
            for t2 in ins[j:e + 1]:
                if t2.startswith(mem + ("s_nop 0x2", "s_waitcnt", "s_barrier")): print("     " + t2)
        if t.startswith("s_nop 0x2b1"):
            e = j
            while not ins[e].startswith(mem): e += 1
            print("  -- wait: " + " ; ".join(x for x in ins[j:e + 4] if x.startswith(mem + ("s_nop 0x2", "s_waitcnt", "v_cmp", "s_cbranch"))))
        if t.startswith("s_nop 0x2c1"):
            e = j
            while not ins[e].startswith("s_nop 0x2c2"): e += 1
            print("  -- DONE word performed before the commit: " + " ; ".join(x for x in ins[j:e + 1] if x.startswith(mem + ("s_nop 0x2", "s_waitcnt", "s_barrier"))))
