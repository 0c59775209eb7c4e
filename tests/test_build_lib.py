"""tools/build_lib.py is the one list of what goes into libhikari_hip.so: every source under csrc/ is on it (a new file that is not
would compile nowhere and fail at link time on somebody else's machine), and the flags keep the numeric contract."""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_lib  # noqa: E402


def test_every_source_is_built_once():
    on_disk = sorted(f for f in os.listdir(build_lib.CSRC) if f.endswith((".hip", ".cpp")))
    assert sorted(build_lib.SOURCES) == on_disk
    assert len(set(build_lib.SOURCES)) == len(build_lib.SOURCES)


def test_flags_keep_the_numeric_contract_and_the_target():
    f = build_lib.FLAGS
    assert "-ffp-contract=off" in f            # only the fmaf() written in the sources fuses
    assert "--offload-arch=gfx950" in f and sum(x.startswith("--offload-arch") for x in f) == 1   # gfx950 only
    assert not any("fast-math" in x or "correctly-rounded" in x for x in f)   # IEEE division / square root stay (profiles/r04_native_division_ab.txt)
