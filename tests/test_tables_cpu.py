"""The constant tables the kernels use (exported by the built library as vvc355_tab_*) against the reference's own files,
without going through ffvvc_amd/csrc/tables.inc / tables_small.inc or their generator: tools/ref_tables.py reads
libavcodec/vvc/vvc_data.c and the inline tables of vvc_filter.c, vvc_intra.c, vvc_intra_template.c and vvc_filter_template.c
(Table 43, the intra angles, the level scale, the filtered modes and filter thresholds, CCLM's divSigTable, ALF's varTab and
transpose index lists, the 4x4 diagonal scan) as data.  Where a kernel uses a packed or arithmetic form of such a table
(intra.hip, itx.hip, alf.hip) a static_assert proves it equal to the exported initialiser at compile time.
Where the reference is present (the build container) the values are compared one by one and the committed digests are checked to
be current; everywhere (the GPU box has no reference) the library's tables are compared with the committed digests
(tests/golden/tables_sha256.json).  The DCT-2 table has no counterpart in vvc_data.c (the reference hard-codes DCT-2 butterflies in
vvc_itx_1d.c); it is checked against the closed form round(64 * sqrt(2) * cos(j * pi / 128))-style magnitudes through the
transform tests instead (tests/test_oracle_cpu.py, tests/test_itx_gpu.py)."""
import ctypes
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_tables  # noqa: E402

CT = {"int8": ctypes.c_int8, "uint8": ctypes.c_uint8, "int16": ctypes.c_int16, "uint16": ctypes.c_uint16}
FMT = {"int8": "b", "uint8": "B", "int16": "h", "uint16": "H"}


def fixture():
    return json.load(open(ref_tables.FIXTURE))["tables"]


def exported(lib, name, meta):
    arr = (CT[meta["type"]] * meta["count"]).in_dll(lib, "vvc355_tab_" + name)
    return list(arr)


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "ffvvc_amd", "libvvc_mi355.so")
    if not os.path.exists(so):
        pytest.fail(f"{so} is missing: build the HIP extension first (__graft_entry__.build())")
    return ctypes.CDLL(so)


@pytest.mark.parametrize("name", sorted(fixture()))
def test_library_table_matches_committed_digest(lib, name):
    meta = fixture()[name]
    got = exported(lib, name, meta)
    assert ref_tables.digest(FMT[meta["type"]], got) == meta["sha256"], f"vvc355_tab_{name} differs from the reference's table"


@pytest.mark.skipif(not os.path.exists(ref_tables.REF), reason="the reference tree is only present in the build container")
def test_against_reference_values_and_fixture_current(lib):
    ref = ref_tables.read_reference()
    fix = fixture()
    assert sorted(ref) == sorted(fix)
    for name, (fmt, want) in ref.items():
        meta = fix[name]
        assert meta["count"] == len(want) and meta["sha256"] == ref_tables.digest(fmt, want), f"tests/golden/tables_sha256.json is stale for {name}"
        got = exported(lib, name, meta)
        bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
        assert not bad, f"vvc355_tab_{name}: {len(bad)} entries differ from vvc_data.c, first at {bad[0]}: {got[bad[0]]} != {want[bad[0]]}"
