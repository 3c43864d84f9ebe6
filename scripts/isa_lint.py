#!/usr/bin/env python3
"""ISA lint of libathena_mp.so's gfx950 code objects (runs in the build container: no GPU needed).

Two things were found by accident in round 4 and cost real time or real bits, with green timings / green builds:
  * a hot kernel silently acquiring a SCRATCH segment (gno_pc_kernel: a compiler-made 96 B/lane array, 5.8 GB of
    traffic per launch, profiles/r04_c4_scratch_ab.txt);
  * the gfx950 data hazard of a buffer store of more than 8 bytes whose scalar offset is a REGISTER, followed at once by a
    VALU write of its data registers (profiles/r04_ubench_store_hazard.txt: LLVM's hazard recogniser pads only the
    immediate-soffset form; the hardware needs two wait states in both).
This tool extracts every gfx950 code object from the shared library (llvm-objdump --offloading), reads the kernel
descriptors' metadata (llvm-readelf --notes) and the disassembly (llvm-objdump -d) and reports per kernel:
  registers / LDS / occupancy, scratch bytes, scratch instructions and how many of them sit INSIDE A LOOP (an address
  interval closed by a backward branch), and every store -> VALU sequence of the hazard's shape.
tests/test_isa_lint.py holds the rules; `python scripts/isa_lint.py --summary profiles/r06_isa_summary.txt` writes the table.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ATHENA_MP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "athena_amd", "libathena_mp.so")

# kernel families on the timed paths of bench.py / scripts/bench_configs.py
TIMED_FAMILIES = ("agg_gemm", "gemm_dw_full", "gno_pc", "gno_dh_pc", "gno_stg", "gno_px_gather", "duv_rows_wide",
                  "duv_bwd_wide", "duv_bwd_ro", "duv_dw_wide", "csr_gather", "readout_fwd", "readout_bwd", "banded_agg_gemm64")
# ... and the exact instantiations the bench lines launch (BASELINE configs[1] .. [4]; BUF = true is what a tensor below 4 GiB takes)
TIMED_EXACT = (
    "agg_gemm_kernel<128, true, 0, true>", "agg_gemm_kernel<128, true, 1, true>", "agg_gemm_kernel<128, false, 0, true>",
    "gemm_dw_full_kernel<128, 128, true>", "gemm_dw_full_kernel<128, 128, false>",
    "agg_gemm256_kernel<true, 0, true>", "agg_gemm256_kernel<true, 0, false>", "agg_gemm256_kernel<false, 0, true>",
    "agg_gemm256_kernel<false, 0, false>",
    "gno_pc_kernel<false>", "gno_pc_kernel<true>", "gno_stg_kernel<false>", "gno_stg_kernel<true>", "gno_px_gather_kernel",
    "gno_dh_pc_kernel<false, 4, 1, true>", "gno_dh_pc_kernel<true, 4, 1, true>", "gno_dh_pc_kernel<false, 2, 2, true>",
    "gno_dh_pc_kernel<true, 2, 2, true>",
    "duv_rows_wide_kernel<5, 4, true, false, 2>", "duv_rows_wide_kernel<5, 4, true, true, 2>", "duv_rows_wide_kernel<5, 4, false, false, 2>", "duv_bwd_wide_kernel<5, 4>",
    "duv_bwd_ro_kernel<5, 2, true>", "duv_bwd_ro_kernel<5, 2, false>",
    "csr_gather_short_rows<16, 4>", "csr_gather_short_rows<64, 4>", "csr_gather_banded64<false>", "csr_gather_banded64<true>",
    "banded_agg_gemm64_kernel<true, 1, false, true>", "banded_agg_gemm64_kernel<false, 0, false, false>", "readout_bwd_kernel<4, 2, false>", "readout_bwd_kernel<4, 2, true>",
)
# Loop-invariant values the register allocator parks across a WHOLE loop nest -- stored once in front of it, reloaded once behind
# it, no scratch instruction inside any loop -- cost nothing and are accepted up to this many bytes on a bench-line kernel
# (gno_pc_kernel: two lane-constant LDS offsets across the long-row tile loop, 12 B at 168 registers = three waves per SIMD).
# Round 4's accident was 96 B per lane read and written INSIDE the tile loop.
PARKED_BYTES_MAX = 16
# Known, measured, not on any bench line: the tanh epilogue (tanhf's range reduction at the 128-register cap of four waves per
# SIMD: two registers parked around it, one reload inside the tile loop).  A ratchet: the byte count may not grow.
KNOWN_SPILLS = {
    "agg_gemm_kernel<64, true, 3, true>": 12, "agg_gemm_kernel<64, true, 3, false>": 12, "agg_gemm_kernel<128, true, 3, false>": 12,
    # 96 inputs + the readout epilogue: two waves per SIMD WITH these spills measured faster than one wave without
    # (0.488 against 0.545 ms, profiles/r05_duv_96wide_ab.txt)
    "duv_rows_wide_kernel<6, 4, true, false, 4>": 44,
}


def tools_available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-readelf")) and shutil.which("c++filt")


def extract(lib, workdir):
    """gfx950 code objects of `lib`, extracted into workdir (llvm-objdump writes next to its INPUT, so the library is
    copied there first)."""
    local = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "hipv4-amdgcn" in f and f.endswith("gfx950"))


_META_KEYS = (".name", ".private_segment_fixed_size", ".sgpr_count", ".sgpr_spill_count", ".vgpr_count", ".vgpr_spill_count",
              ".group_segment_fixed_size", ".max_flat_workgroup_size", ".uses_dynamic_stack")


def metadata(code_object):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code_object], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        d = {"agpr": int(block.split("\n")[0].strip())}
        for k in _META_KEYS:
            m = re.search(r"\n\s+" + re.escape(k) + r":\s+(\S+)", block)
            d[k.lstrip(".")] = m.group(1) if m else None
        out[d["name"]] = d
    return out


def demangle(names):
    res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    short = []
    for n in res[:len(names)]:
        n = n.replace("(anonymous namespace)::", "")
        n = re.sub(r"^void ", "", n)
        depth, cut = 0, len(n)      # cut the argument list: the first '(' outside template brackets
        for i, ch in enumerate(n):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        short.append(n[:cut])
    return short


_FUNC = re.compile(r"^([0-9a-f]{16}) <(\S+)>:")
_INSN = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<(\S+?)\+0x([0-9a-f]+)>\s*$")


def disassemble(code_object):
    """{mangled kernel name: [(address, mnemonic, operand string, branch target or None)]}"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", code_object], check=True,
                         capture_output=True, text=True).stdout
    funcs, cur, base = {}, None, 0
    for line in txt.split("\n"):
        m = _FUNC.match(line)
        if m:
            base, cur = int(m.group(1), 16), []
            funcs[m.group(2)] = cur
            continue
        if cur is None:
            continue
        m = _INSN.match(line)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = _TARGET.search(line)
            if t:
                tgt = base + int(t.group(2), 16)
        cur.append((addr, mnem, ops, tgt))
    return funcs


def _vregs(tok):
    """register numbers of a VGPR operand token ('v12', 'v[4:7]'); empty for anything else"""
    tok = tok.strip()
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _split_ops(ops):
    out, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


_WIDE_STORE = re.compile(r"^buffer_store_(dwordx3|dwordx4|b96|b128)$")


def store_hazards(insns, wait_states=2):
    """buffer stores of more than 8 bytes with the scalar offset in an SGPR, followed within `wait_states` wait states by a
    VALU instruction that writes one of the store's data registers (profiles/r04_ubench_store_hazard.txt, case A).
    s_nop N counts N + 1 wait states, any other instruction one."""
    found = []
    for i, (addr, mnem, ops, _) in enumerate(insns):
        if not _WIDE_STORE.match(mnem):
            continue
        o = _split_ops(ops)
        if len(o) < 4:
            continue
        data = _vregs(o[0])
        soff = o[3].split()[0] if o[3] else ""
        if not re.fullmatch(r"s\d+|s\[\d+:\d+\]|m0", soff):
            continue                      # immediate / literal soffset: LLVM's own hazard padding applies
        waited, j = 0, i + 1
        while j < len(insns) and waited < wait_states:
            a2, m2, o2, _ = insns[j]
            if m2 == "s_nop":
                try:
                    waited += int(o2.split()[0], 0) + 1
                except ValueError:
                    waited += 1
                j += 1
                continue
            if m2.startswith("v_"):
                dst = _split_ops(o2)
                if dst and (_vregs(dst[0]) & data):
                    found.append((addr, f"{mnem} {ops}", a2, f"{m2} {o2}"))
                    break
            waited += 1
            j += 1
    return found


def loops(insns):
    """[(first address, last address)] of every interval closed by a backward branch"""
    return [(tgt, addr) for addr, mnem, _, tgt in insns if tgt is not None and tgt <= addr]


def occupancy(meta, waves_per_wg):
    """waves per SIMD the register file and the LDS allow (gfx950: 512 unified registers per lane and SIMD, allocated in
    blocks of 8; 160 KB of LDS per CU; at most 8 waves per SIMD)"""
    regs = int(meta["vgpr_count"]) if int(meta["agpr"]) == 0 else ((int(meta["vgpr_count"]) + 7) // 8) * 8
    regs = max(8, ((regs + 7) // 8) * 8)
    by_regs = min(8, 512 // regs)
    lds = int(meta["group_segment_fixed_size"])
    if lds > 0 and waves_per_wg > 0:
        wgs = (160 * 1024) // lds
        by_lds = max(0, (wgs * waves_per_wg) // 4)
        return min(by_regs, by_lds) if by_lds else by_regs
    return by_regs


def analyse(lib=DEFAULT_LIB):
    """[{name, short, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, occupancy, scratch_insns, scratch_in_loop,
        hazards}] for every kernel of the library"""
    rows = []
    with tempfile.TemporaryDirectory(prefix="isa_lint_") as wd:
        for co in extract(lib, wd):
            meta = metadata(co)
            funcs = disassemble(co)
            for name, m in meta.items():
                insns = funcs.get(name, [])
                lp = loops(insns)
                scr = [(a, mn) for a, mn, _, _ in insns if mn.startswith("scratch_")]
                in_loop = [a for a, _ in scr if any(t <= a <= b for t, b in lp)]
                wg = int(m["max_flat_workgroup_size"] or 256)
                rows.append({
                    "name": name, "vgpr": int(m["vgpr_count"]), "agpr": int(m["agpr"]), "sgpr": int(m["sgpr_count"]),
                    "vgpr_spill": int(m["vgpr_spill_count"]), "sgpr_spill": int(m["sgpr_spill_count"]),
                    "scratch": int(m["private_segment_fixed_size"]), "lds": int(m["group_segment_fixed_size"]),
                    "dynamic_stack": m["uses_dynamic_stack"] == "true", "wg": wg, "occupancy": occupancy(m, (wg + 63) // 64),
                    "n_insns": len(insns), "scratch_insns": len(scr), "scratch_in_loop": len(in_loop),
                    "hazards": store_hazards(insns),
                })
    for r, s in zip(rows, demangle([r["name"] for r in rows])):
        r["short"] = s
    return rows


def family_of(short):
    for f in TIMED_FAMILIES:
        if short.startswith(f):
            return f
    return None


def violations(rows, require_exact=True):
    """The rules (tests/test_isa_lint.py):
       R0  every instantiation named in TIMED_EXACT exists (a rename must update the list, or the lint guards nothing)
       R1  an instantiation a bench line launches executes NO scratch instruction inside a loop, has no dynamic stack and at most
           PARKED_BYTES_MAX bytes of scratch (see there)
       R2  no other kernel of a timed family executes a scratch instruction inside a loop, allocates a dynamic stack or has more
           than 128 bytes of scratch -- except the instantiations of KNOWN_SPILLS, whose byte count may not grow
       R3  no kernel of the library has the store -> VALU sequence of the gfx950 hazard"""
    out = []
    exact_seen = set()
    for r in rows:
        fam = family_of(r["short"])
        if r["short"] in TIMED_EXACT:
            exact_seen.add(r["short"])
            if r["scratch_in_loop"] or r["dynamic_stack"] or r["scratch"] > PARKED_BYTES_MAX:
                out.append(f"R1 {r['short']}: {r['scratch']} B of scratch, {r['scratch_in_loop']} scratch instruction(s) inside a loop "
                           f"(vgpr spills {r['vgpr_spill']}, sgpr spills {r['sgpr_spill']})")
        elif r["short"] in KNOWN_SPILLS:
            if r["scratch"] > KNOWN_SPILLS[r["short"]] or r["dynamic_stack"]:
                out.append(f"R2 {r['short']}: {r['scratch']} B of scratch, recorded {KNOWN_SPILLS[r['short']]} B")
        elif fam:
            if r["scratch_in_loop"]:
                out.append(f"R2 {r['short']}: {r['scratch_in_loop']} scratch instruction(s) inside a loop")
            if r["dynamic_stack"]:
                out.append(f"R2 {r['short']}: dynamic stack")
            if r["scratch"] > 128:
                out.append(f"R2 {r['short']}: {r['scratch']} B of scratch")
        for h in r["hazards"]:
            out.append(f"R3 {r['short']}: {h[1]} @{h[0]:x} then {h[3]} @{h[2]:x}")
    if require_exact:
        for e in TIMED_EXACT:
            if e not in exact_seen:
                out.append(f"R0 {e}: instantiation not found in the library (rename? update TIMED_EXACT)")
    return out


def summary(rows, only_timed=True):
    lines = ["# kernel | vgpr agpr sgpr | spills v/s | scratch B (insns, in loop) | LDS B | waves/SIMD | insns | hazards"]
    for r in sorted(rows, key=lambda r: r["short"]):
        if only_timed and not family_of(r["short"]):
            continue
        mark = " *" if r["short"] in TIMED_EXACT else ""
        lines.append(f"{r['short'] + mark:58s} | {r['vgpr']:3d} {r['agpr']:3d} {r['sgpr']:3d} | {r['vgpr_spill']:3d}/{r['sgpr_spill']:3d} | "
                     f"{r['scratch']:4d} ({r['scratch_insns']}, {r['scratch_in_loop']}) | {r['lds']:6d} | {r['occupancy']} | {r['n_insns']:6d} | {len(r['hazards'])}")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--summary", help="write the per-kernel table here")
    ap.add_argument("--all", action="store_true", help="every kernel, not only the timed families")
    a = ap.parse_args()
    if not tools_available():
        print("llvm-objdump / llvm-readelf / c++filt not found")
        return 3
    rows = analyse(a.lib)
    text = summary(rows, only_timed=not a.all)
    v = violations(rows)
    text += ("\n# * = an instantiation a bench line launches (R1: no scratch instruction inside a loop, at most "
             f"{PARKED_BYTES_MAX} B parked across a loop nest)\n# violations: ") + (str(len(v)) if v else "none") + "\n"
    text += "".join(f"#   {x}\n" for x in v)
    if a.summary:
        with open(a.summary, "w") as f:
            f.write(f"# scripts/isa_lint.py on {os.path.relpath(a.lib, ROOT)} ({len(rows)} kernels in the library)\n" + text)
    print(text)
    return 1 if v else 0


if __name__ == "__main__":
    sys.exit(main())
