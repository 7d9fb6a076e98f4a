"""Mutation testing of the CPU oracle: which of the reference's goldens pins which reading of the source?

oracle/rogue_oracle.c carries one switchable misreading per RNG call site of SURVEY.md App. B (32 <-> 64-bit sample width, draw order, range bound) and per
quirk of App. C (-DORC_MUTANT=k; tests/oracle_mutants.py lists them).  Every mutant is built and run against the checks of tests/test_oracle_golden.py; the
result must equal the committed pin map (tests/golden/mutant_pins.json, regenerated with profiles/r05_pin_map.txt by tools/pin_map.py):
  * a mutant noticed by a REFERENCE golden: that reading is pinned by data the reference itself holds;
  * the others are listed as unpinned by the reference -- there parity rests on the source text, and tests/test_oracle_shadow.py re-derives those
    functions from the text a second time.
CPU only (8 s on 8 cores)."""
import json
import os
import re

from oracle_mutants import MUTANTS, REFERENCE, SECONDARY, kill_matrix, summarise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_mutant_site_exists_in_the_oracle_source():
    src = open(os.path.join(ROOT, "oracle", "rogue_oracle.c")).read()
    used = {int(k) for k in re.findall(r"\b(?:MUT|W32|W64|DH|PC)\((\d+)", src)}
    assert used == set(MUTANTS), (sorted(used - set(MUTANTS)), sorted(set(MUTANTS) - used))


def test_pin_map_matches_the_committed_one():
    pins = summarise(kill_matrix())
    assert pins["0"] == {"reference": [], "secondary": []}, "the restatement itself fails a golden"
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "mutant_pins.json")))
    assert pins == want, {k: (pins.get(k), want.get(k)) for k in set(pins) | set(want) if pins.get(k) != want.get(k)}
    killed = [k for k in MUTANTS if pins[str(k)]["reference"]]
    assert len(killed) >= 35  # every generator call site but two (M2, M28) and the corridor paint rule (M45) is pinned by a reference golden
    # the published map names every survivor
    text = open(os.path.join(ROOT, "profiles", "r05_pin_map.txt")).read()
    for k in MUTANTS:
        assert re.search(r"^M%d\s" % k, text, re.M), "M%d is missing from profiles/r05_pin_map.txt" % k
    unpinned = text.split("== NOTICED ONLY BY SECONDARY FIXTURES")[1]
    for k in MUTANTS:
        if not pins[str(k)]["reference"]:
            assert re.search(r"^M%d\s" % k, unpinned, re.M), "M%d is unpinned by the reference but not listed as such" % k


def test_choose_width_mutant_is_the_one_the_golden_test_names():
    """tests/test_oracle_golden.py::test_choose_is_64bit is mutant 20 by another route (a run-time switch): both must agree."""
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "mutant_pins.json")))
    assert "seed1_clear_map" in want["20"]["reference"]
    assert set(REFERENCE).isdisjoint(SECONDARY)
