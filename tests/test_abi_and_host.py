"""CPU: the C-ABI library loads and exports every symbol include/trajadmm.h declares, fails loudly
without a GPU (no CPU fallback), and the host-side pieces (scene generators, file formats, CLIs'
argument/config handling) behave like the reference's."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _header_functions(name="trajadmm.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tj_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    assert os.path.exists(pkg.LIB_PATH), "libtrajadmm.so missing: run __graft_entry__.build()"
    lib = C.CDLL(pkg.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/trajadmm.h but not exported"
    assert set(pkg.EXPORTS) <= set(declared)
    # the product library carries no test surface; the known-answer hooks (include/trajadmm_kat.h) are exported by the TEST build only
    kat_declared = [n for n in _header_functions("trajadmm_kat.h") if n.startswith("tj_kat_")]
    assert sorted(kat_declared) == sorted(pkg.KAT_EXPORTS)
    assert not [n for n in kat_declared if hasattr(lib, n)]
    assert os.path.exists(pkg.KAT_LIB_PATH), "libtrajadmm_kat.so missing: run __graft_entry__.build()"
    kat = C.CDLL(pkg.KAT_LIB_PATH)
    for name in declared + kat_declared:
        assert hasattr(kat, name), f"{name} not exported by the test build"


def test_params_struct_layout_matches_header(pkg):
    """ctypes mirror and C struct must agree: tj_default_params fills the shipped 3D.json values"""
    lib = C.CDLL(pkg.LIB_PATH)
    p = pkg.TjParams()
    lib.tj_default_params(C.byref(p), 1, 64, 5)
    assert (p.mode, p.uav_num, p.piece_num, p.res) == (1, 64, 5, 8)
    assert (p.lambda_, p.margin, p.offset, p.mu, p.vel_limit, p.acc_limit) == (10.0, 0.1, 0.1, 0.1, 2.0, 2.0)
    assert (p.ks, p.kt, p.stop) == (1e-3, 1.0, 1e-2) and (p.rank, p.world) == (0, 1)
    lib.tj_default_params(C.byref(p), 0, 1, 5)
    assert p.ks == 1e-8


def test_no_cpu_fallback(pkg, scenes):
    """without a HIP device the product must refuse to run rather than compute on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.TrajAdmmError) as ei:
        pkg.Solver(scenes.tiny(1))
    assert "-2" in str(ei.value) or "HIP" in str(ei.value) or "hip" in str(ei.value)


def test_invalid_params_rejected(pkg):
    lib = C.CDLL(pkg.LIB_PATH)
    lib.tj_last_error.restype = C.c_char_p
    p = pkg.TjParams()
    lib.tj_default_params(C.byref(p), 0, 3, 5)        # single mode with 3 robots
    ctx = C.c_void_p()
    assert lib.tj_create(C.byref(p), C.byref(ctx)) == -1
    assert b"SINGLE" in lib.tj_last_error(ctx)
    lib.tj_destroy(ctx)
    lib.tj_default_params(C.byref(p), 1, 4, 1)        # fewer than 2 pieces
    assert lib.tj_create(C.byref(p), C.byref(ctx)) == -1
    lib.tj_destroy(ctx)


def test_scenes_are_deterministic(scenes):
    a, b = scenes.scn_b(), scenes.scn_b()
    assert np.array_equal(a["cloud"], b["cloud"]) and np.array_equal(a["waypoints"], b["waypoints"])
    c = scenes.scn_c()
    assert c["U"] == 64 and c["cloud"].shape == (100000, 3) and c["P"] == 5 and c["mode"] == 1
    s = scenes.scn_a()
    assert s["mode"] == 0 and s["U"] == 1 and s["ks"] == 1e-8
    # tube is free of points
    assert ((s["cloud"][:, 1] - 0.8 * np.sin(s["cloud"][:, 0])) ** 2 + s["cloud"][:, 2] ** 2 > 0.45 ** 2).all()


def test_reference_file_formats_roundtrip(scenes, tmp_path):
    sc = scenes.tiny(1)
    scenes.write_reference_files(sc, str(tmp_path), "t.obj")
    lines = open(tmp_path / "model" / "multiple" / "t.obj").read().split("\n")
    assert lines[0].startswith("v ") and len([l for l in lines if l.startswith("v ")]) == sc["cloud"].shape[0]
    rows = [l.split() for l in open(tmp_path / "init" / "t.obj_init_file.txt").read().strip().split("\n")]
    assert len(rows) == sc["P"] + 1 and len(rows[0]) == 3 * sc["U"]     # 3*U numbers per line (multiPathPlanning3D.cpp:89-110)
    back = np.array(rows, dtype=float).reshape(sc["P"] + 1, sc["U"], 3).transpose(1, 0, 2) * 5
    assert np.allclose(back, sc["waypoints"], rtol=1e-15)


def _cli(name):
    return os.path.join(ROOT, "traj-opt-admm_amd", name)


@pytest.mark.parametrize("name", ["admmPathPlanning3D", "multiPathPlanning3D"])
def test_cli_usage_and_config_errors(name, tmp_path):
    exe = _cli(name)
    assert os.path.exists(exe), "CLI not built: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "Syntax" in r.stderr                  # same usage line as the reference mains
    r = subprocess.run([exe, "x.obj"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "Config_File/3D.json" in r.stderr     # CWD-relative config path, underscore spelling
    os.makedirs(tmp_path / "Config_File")
    (tmp_path / "Config_File" / "3D.json").write_text('{"auto":0,"init":1,"gui":0}')
    r = subprocess.run([exe, "x.obj"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "missing key" in r.stderr             # all 16 keys are mandatory
    shipped = '{"auto":0,"init":1,"gui":1,"optimal_plane":0,"decouple":1,"res":8,"vel_limit":2,"acc_limit":2,"lambda":1e1,' \
              '"epsilon":1e-1,"margin":1e-1,"offset":1e-1,"stop":1e-2,"exit":0,"init_ob":1,"mu":0.1}'
    (tmp_path / "Config_File" / "3D.json").write_text(shipped)
    r = subprocess.run([exe, "x.obj"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "gui:1" in r.stderr                   # unsupported branch is rejected, not ignored


@pytest.mark.parametrize("P", [2, 5])
def test_host_tables_bit_exact_vs_reference(pkg, P):
    """the library's own host precompute (tj_host_tables, no GPU needed) against the tables the unmodified
    reference produced: C2 junction maps, jerk Gram matrix, blossom subdivision bases, 49 k-DOP axes"""
    g = np.load(os.path.join(ROOT, "tests", "golden", f"tables_P{P}.npz"))
    conv, M, basis, kdop = pkg.host_tables(P, 8)
    assert np.array_equal(conv, g["convert"])
    assert np.array_equal(M, g["mdyn"])
    assert np.array_equal(basis, g["basis"])
    assert np.array_equal(kdop, g["kdop"])


def test_coupled_mode_params(pkg):
    lib = C.CDLL(pkg.LIB_PATH)
    lib.tj_last_error.restype = C.c_char_p
    p = pkg.TjParams()
    lib.tj_default_params(C.byref(p), 2, 8, 5)        # TJ_MODE_MULTI_COUPLED keeps the multi main's ks
    assert p.mode == 2 and p.ks == 1e-3
    p.world, p.rank = 2, 2                            # argument validation happens before any device work: rank must be < world
    ctx = C.c_void_p()
    assert lib.tj_create(C.byref(p), C.byref(ctx)) == -1
    assert b"rank" in lib.tj_last_error(ctx)
    lib.tj_destroy(ctx)
    p.world, p.rank = 2, 0                            # coupled mode shards across ranks since round 2: on a box without a GPU the only
    ctx = C.c_void_p()                                # possible failure is the missing device (there is no CPU fallback)
    assert lib.tj_create(C.byref(p), C.byref(ctx)) in (0, -2)
    lib.tj_destroy(ctx)


def test_hot_kernels_use_no_scratch_memory(pkg, tmp_path):
    """The kernels of the iteration chain whose critical path is a per-item dependent chain must not touch scratch (private)
    memory: every access is a global-memory round trip on that chain.  (Round 2 found k_grad's running sums there -- handed
    around by reference, the compiler had turned them into an indexed array: 1.5 us and 4 MB of writes per launch -- and the
    slack body's scalar Armijo loop reloading 35 spilled registers per trial.)  Read from the code object's own metadata;
    k_ccd_lean spills by design (its occupancy cap), the debug / planner hooks are not on the path."""
    readelf, objdump = "/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(readelf) and os.path.exists(objdump)):
        pytest.skip("LLVM binutils of the ROCm image not found")
    so = tmp_path / "lib.so"
    so.write_bytes(open(pkg.LIB_PATH, "rb").read())
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    cos = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert cos, "no gfx950 code object embedded in libtrajadmm.so"
    notes = subprocess.run([readelf, "--notes", str(tmp_path / cos[0])], check=True, capture_output=True, text=True).stdout
    name = None
    seen = {}
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
        if m and name:
            seen[name] = int(m.group(1))
    hot = [k for k in seen if re.search(r"tj\d+(k_grad|k_xsolve|k_xsolve_band|k_linesearch|k_front|k_mid|k_slack|k_ccd_lean)(I|E)", k)]   # k_ccd_lean: the chain's build of k_ccd (round 5: no spills)
    assert len(hot) >= 10, f"expected the chain's kernels in the metadata, found {sorted(seen)[:5]}..."
    bad = {k: seen[k] for k in hot if seen[k] != 0}
    if bad:
        # A private segment that is DECLARED but never touched (round 4: k_linesearch, 68 bytes -- the frame objects of its SGPR spills, which all
        # go to VGPR lanes) costs the dispatch ~0.2 us (tools/micro/wb_probe.hip) but no round trip on any chain: allowed when the kernel's own
        # instructions hold no access to it.
        dis = subprocess.run([objdump, "-d", str(tmp_path / cos[0])], check=True, capture_output=True, text=True).stdout
        body, cur = {}, None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1); body[cur] = []
            elif cur:
                body[cur].append(line)
        # (the parse is checked on every kernel looked at: the first token of a line must be the mnemonic -- a body without s_endpgm or global_load would mean the
        #  filter below matches nothing and passes vacuously.)  Exemptions are per kernel: k_linesearch's declared-but-untouched frame; k_ccd_lean's finisher keeps the
        #  stack of the reference's tree query for the order-dependent pair replay in private memory (a cold path: taken when two acting pairs of a segment share a robot)
        allow_touch = {"k_ccd_lean": 128}
        for k in list(bad):
            insts = [l.split()[0] for l in body.get(k, []) if l.strip()]
            assert insts, f"{k} not found in the disassembly"
            assert "s_endpgm" in insts and any(i.startswith("global_load") for i in insts), f"disassembly of {k} not parsed into mnemonics"
            touched = any(i.startswith(("scratch_", "buffer_load", "buffer_store")) for i in insts)
            cold = next((lim for nm, lim in allow_touch.items() if nm in k), 0)
            if bad[k] <= 128 and (not touched or bad[k] <= cold):
                del bad[k]
    assert not bad, f"scratch memory in hot kernels: {bad}"


def _disassemble(pkg, tmp_path):
    """{kernel symbol: [(mnemonic, operand text)]} of the gfx950 code object inside the shipped library"""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("LLVM binutils of the ROCm image not found")
    so = tmp_path / "lib.so"
    so.write_bytes(open(pkg.LIB_PATH, "rb").read())
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    cos = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert cos, "no gfx950 code object embedded in libtrajadmm.so"
    dis = subprocess.run([objdump, "-d", str(tmp_path / cos[0])], check=True, capture_output=True, text=True).stdout
    body, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1); body[cur] = []
        elif cur and line.strip():
            text = line.split("//")[0].strip()
            if text:
                parts = text.split(None, 1)
                body[cur].append((parts[0], parts[1] if len(parts) > 1 else ""))
    return body


def test_signalling_sites_keep_their_order(pkg, tmp_path):
    """Every cross-block / cross-queue protocol of the library ("Synchronisation protocols") rests on one idiom: records out with write-through
    stores, `s_waitcnt vmcnt(0)` (+ a barrier where several waves stored), THEN the signal -- a relaxed atomic; the consumer polls the word and reads afterwards.  Relaxed
    atomics promise no such order in the HIP memory model; the instruction stream does.  This test reads the shipped code object and fails if a compiler change takes the
    order away.  The sites carry `s_nop` immediates no compiler emits (dev_common.h: sig_acked 0x2a1 / sig_sent 0x2a2, wait_begin 0x2b1 / wait_end 0x2b2, and 0x2c1 / 0x2c2
    around "the DONE word is performed before the commit stores" in k_linesearch):
      * right behind 0x2a1 stands the full `s_waitcnt vmcnt(0) ...`; from there to 0x2a2 there is no memory write but the signal itself -- a 32-bit or 64-bit atomic, or
        a single 32-bit write-through store, or ONE 64-bit write-through store on its own (a head-start entry's tag); record stores are 64 bits wide and come in
        numbers: one sunk below the wait would show here;
      * the first memory instruction behind 0x2b1 is the poll itself, a 32- or 64-bit load past the caches (sc1) -- a record load hoisted to the head of the wait would
        stand there instead (the block layout between the loop's markers is not its control flow, so the loop body is not scanned further);
      * between 0x2c1 and 0x2c2 stand the wait and the barrier and no store at all."""
    body = _disassemble(pkg, tmp_path)
    chain = [k for k in body if re.search(r"tj\d+(k_grad|k_xsolve|k_linesearch|k_front|k_mid|k_ccd_lean|k_ccd|k_keep|k_begin|k_xs_gate|k_fa_gate|k_keep_gate|k_ls_coupled)(I|E)", k)]
    assert len(chain) >= 12, sorted(body)[:5]
    n_sig = n_wait = n_word = 0
    writes = ("global_store", "flat_store", "scratch_store", "buffer_store", "global_atomic", "flat_atomic", "buffer_atomic")
    for k in chain:
        ins = body[k]
        assert any(m == "s_endpgm" for m, _ in ins) and any(m.startswith("global_load") for m, _ in ins), f"disassembly of {k} not parsed into mnemonics"
        i = 0
        while i < len(ins):
            m, ops = ins[i]
            if m == "s_nop" and ops.strip() in ("0x2a1", "673"):
                n_sig += 1
                assert ins[i + 1][0] == "s_waitcnt" and "vmcnt(0)" in ins[i + 1][1], f"{k}: no s_waitcnt vmcnt(0) right behind the 'acknowledged' marker: {ins[i + 1]}"
                j = i + 2
                wide = other = 0
                while j < len(ins) and not (ins[j][0] == "s_nop" and ins[j][1].strip() in ("0x2a2", "674")) and ins[j][0] != "s_endpgm":
                    mm, oo = ins[j]
                    if mm.startswith(writes):
                        if mm == "global_store_dwordx2" and "sc1" in oo:
                            wide += 1      # a 64-bit word that IS the signal (a head-start entry's tag); allowed only alone
                        else:
                            other += 1
                            is_signal = "atomic" in mm or (mm == "global_store_dword" and "sc1" in oo)
                            assert is_signal, f"{k}: a store between 'acknowledged' and the signal: {mm} {oo}"
                    j += 1
                assert wide == 0 or (wide == 1 and other == 0), f"{k}: 64-bit stores between 'acknowledged' and the signal ({wide} of them, {other} other writes): a record store sunk below the wait?"
                assert j < len(ins) and ins[j][0] == "s_nop", f"{k}: 'acknowledged' marker without a 'sent' marker behind it"
                i = j
            elif m == "s_nop" and ops.strip() in ("0x2b1", "689"):
                n_wait += 1
                j = i + 1
                while not ins[j][0].startswith(("global_", "flat_", "buffer_", "scratch_")):   # the first memory instruction behind the marker, in layout order, is the poll
                    assert ins[j][0] != "s_endpgm", f"{k}: a wait without a poll"
                    j += 1
                assert ins[j][0] in ("global_load_dword", "global_load_dwordx2") and "sc1" in ins[j][1], f"{k}: the first memory access of a wait is not a poll past the caches: {ins[j]}"
            elif m == "s_nop" and ops.strip() in ("0x2c1", "705"):
                n_word += 1
                j = i + 1
                seen_wait = seen_barrier = False
                while not (ins[j][0] == "s_nop" and ins[j][1].strip() in ("0x2c2", "706")):
                    mm, oo = ins[j]
                    seen_wait |= mm == "s_waitcnt" and "vmcnt(0)" in oo
                    seen_barrier |= mm == "s_barrier"
                    assert not mm.startswith(writes), f"{k}: a memory write between the DONE word's wait and the commit: {mm} {oo}"
                    j += 1
                assert seen_wait and seen_barrier, f"{k}: the DONE word's wait / barrier is gone"
                i = j
            i += 1
    # the sites exist (a refactoring that drops the markers must not turn this test into a no-op): producers in k_grad, k_xsolve, k_linesearch, k_front, k_ccd, k_keep, k_mid's watcher, ...
    assert n_sig >= 20 and n_wait >= 20 and n_word >= 1, (n_sig, n_wait, n_word)


def test_every_environment_switch_is_documented():
    """Every environment switch the library reads goes through tune("KEY") (csrc/tj_api.hip; TJ_TUNE="KEY=value,..." or TJ_KEY=value) -- no bare getenv of a TJ_ variable
    anywhere in csrc -- and every KEY is listed in INTEGRATION.md's table; the table lists no switch the source no longer reads."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ""
    for f in sorted(os.listdir(os.path.join(root, "traj-opt-admm_amd", "csrc"))):
        if f.endswith((".h", ".hip", ".cpp")):
            src += open(os.path.join(root, "traj-opt-admm_amd", "csrc", f)).read()
    bare = set(re.findall(r'getenv\("(TJ_[A-Z0-9_]+)"\)', src)) - {"TJ_TUNE"}
    assert not bare, f"environment switches read outside tune(): {sorted(bare)}"
    keys = set(re.findall(r'tune\("([A-Z0-9_]+)"\)', src))
    assert len(keys) >= 30
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    table = doc[doc.index("### Environment switches"):doc.index("## 3. Python")]
    listed = set(re.findall(r"`TJ_([A-Z0-9_]+)", table)) - {"TUNE", "KEY", "PRINT_OBSERVED", "ERR_NO_PROGRESS", "ERR_UNSUPPORTED", "LS_EXACT_HULLS", "LS_HELP_FORCE", "SHARD_UNFUSED"}
    assert keys - listed == set(), f"switches missing from INTEGRATION.md: {sorted(keys - listed)}"
    assert listed - keys == set(), f"INTEGRATION.md lists switches the source does not read: {sorted(listed - keys)}"


def test_rccl_entry_points_resolve(pkg):
    """the "rccl" transport of tj_group binds librccl.so at run time (dlopen): the library must open here and export
    ncclCommInitAll / ncclAllGather / ncclCommDestroy / ncclGetErrorString -- the link step of that transport, which needs no GPU"""
    import ctypes
    lib = ctypes.CDLL(pkg.LIB_PATH)
    assert lib.tj_rccl_available() == 1
