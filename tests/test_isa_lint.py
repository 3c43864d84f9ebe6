"""ISA lint of the shipped library (scripts/isa_lint.py): no scratch inside the loops of the timed kernels, no gfx950
store -> VALU hazard anywhere.  Build container only (cross-compiles and disassembles; no GPU).  VERDICT r04 item 3."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_lint  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not (isa_lint.tools_available() and os.path.exists(HIPCC)),
                                reason="needs hipcc, llvm-objdump, llvm-readelf, c++filt (the build container)")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


@pytest.fixture(scope="module")
def shipped():
    lib = os.path.join(ROOT, "athena_amd", "libathena_mp.so")
    if not os.path.exists(lib):
        subprocess.check_call(["bash", os.path.join(ROOT, "athena_amd", "csrc", "build.sh")])
    return isa_lint.analyse(lib)


def test_shipped_library_is_clean(shipped):
    v = isa_lint.violations(shipped)
    assert not v, "\n".join(v)
    by = {r["short"]: r for r in shipped}
    for name in isa_lint.TIMED_EXACT:            # every bench-line instantiation was found and looked at
        assert by[name]["n_insns"] > 50, name
        assert by[name]["scratch_in_loop"] == 0
    # the headline kernels carry no scratch segment at all
    for name in ("agg_gemm_kernel<128, true, 0, true>", "gemm_dw_full_kernel<128, 128, true>", "gemm_dw_full_kernel<128, 128, false>",
                 "duv_rows_wide_kernel<5, 4, true, false, 2>", "duv_rows_wide_kernel<5, 4, true, true, 2>", "duv_bwd_wide_kernel<5, 4>", "duv_bwd_ro_kernel<5, 2, true>", "gno_stg_kernel<false>", "gno_px_gather_kernel"):
        assert by[name]["scratch"] == 0, (name, by[name]["scratch"])


def test_summary_in_profiles_is_current(shipped):
    """profiles/r06_isa_summary.txt is the table of THIS library (registers / occupancy per timed kernel)"""
    path = os.path.join(ROOT, "profiles", "r06_isa_summary.txt")
    assert os.path.exists(path), "run: python scripts/isa_lint.py --summary profiles/r06_isa_summary.txt"
    have = {}
    for line in open(path):
        if line.startswith("#") or "|" not in line:
            continue
        f = [x.strip() for x in line.split("|")]
        have[f[0].rstrip(" *")] = (f[1].split(), f[3].split()[0])
    for r in shipped:
        if isa_lint.family_of(r["short"]) is None:
            continue
        assert r["short"] in have, r["short"]
        regs, scratch = have[r["short"]]
        assert [int(x) for x in regs] == [r["vgpr"], r["agpr"], r["sgpr"]], r["short"]
        assert int(scratch) == r["scratch"], r["short"]


def test_lint_flags_the_known_bad_shapes(tmp_path):
    """a per-lane indexed private array inside a loop (R2) and the store -> VALU sequence (R3) are reported; the padded
    sequence is not"""
    lib = tmp_path / "libisa_lint_bad.so"
    subprocess.check_call([HIPCC, *FLAGS, "-shared", os.path.join(ROOT, "tests", "fixtures", "isa_lint_bad.hip"), "-o", str(lib)],
                          stderr=subprocess.DEVNULL)
    rows = {r["short"]: r for r in isa_lint.analyse(str(lib))}
    arr = rows["agg_gemm_fixture_indexed_array"]
    assert arr["scratch"] > 128 and arr["scratch_in_loop"] >= 1
    assert len(rows["gno_pc_fixture_hazard"]["hazards"]) == 1
    assert rows["gno_pc_fixture_padded"]["hazards"] == []
    v = isa_lint.violations(list(rows.values()), require_exact=False)
    assert any(x.startswith("R2 agg_gemm_fixture_indexed_array") for x in v)
    assert any(x.startswith("R3 gno_pc_fixture_hazard") for x in v)
    assert not any("padded" in x for x in v)


def test_lint_goes_red_when_by_group_is_reverted(tmp_path):
    """Round 4's accident, replayed: GnoProd::by_group written as the indexed read a[g] makes the compiler park the array in
    scratch and read it back with a per-lane index inside the tile loop (5.8 GB per launch at configs[3],
    profiles/r04_c4_scratch_ab.txt).  The lint must see it in gno_pc_kernel."""
    src = open(os.path.join(ROOT, "athena_amd", "csrc", "gno.hip")).read()
    a = src.index("    __device__ __forceinline__ int by_group(const int (&a)[4]) const")
    b = src.index("    __device__ __forceinline__ void ids_entries(GnoIds &I) const")
    reverted = src[:a] + "    __device__ __forceinline__ int by_group(const int (&a)[4]) const { return a[g]; }\n" + src[b:]
    (tmp_path / "gno.hip").write_text(reverted)
    lib = tmp_path / "libgno_reverted.so"
    subprocess.check_call([HIPCC, *FLAGS, "-I", os.path.join(ROOT, "athena_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-shared", str(tmp_path / "gno.hip"), "-o", str(lib)], stderr=subprocess.DEVNULL)
    rows = isa_lint.analyse(str(lib))
    v = isa_lint.violations(rows, require_exact=False)
    assert any(x.startswith("R1 gno_pc_kernel<false>") for x in v), v
    pc = next(r for r in rows if r["short"] == "gno_pc_kernel<false>")
    assert pc["scratch_in_loop"] > 0 and pc["scratch"] > isa_lint.PARKED_BYTES_MAX
