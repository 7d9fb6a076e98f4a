"""Runs checks of tests/test_oracle_golden.py against whatever oracle library ROGUE_ORACLE_SO names (a mutant build of oracle/rogue_oracle.c) and prints
one `PROBE <check> <0|1>` line per check, flushed as it goes -- a mutant may trip one of the oracle's own aborts, and the parent (tests/oracle_mutants.py
run_mutant) then knows which check it died in.  A child process: the oracle binding loads one library per process.  argv: the checks to run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_oracle_golden as T
    from oracle import pyoracle
    goldens = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))
    print("PROBE mutant %d" % int(pyoracle.lib().orc_mutant_id()), flush=True)
    for name in sys.argv[1:]:
        print("PROBE-START %s" % name, flush=True)
        base, _, arg = name.partition("[")
        fn = getattr(T, "test_" + base)
        try:
            if base == "xorshift_known_answers":
                fn()
            elif arg:
                keys = arg.rstrip("]")
                fn(goldens, keys, {"CMD_STR": "SEED1_DUNGEON2", "CMD_STR5": "SEED1_DUNGEON3"}[keys])
            else:
                fn(goldens)
            ok = 1
        except BaseException:  # noqa: BLE001  (an assertion, a RuntimeError of the binding, pytest.fail's exception)
            ok = 0
        print("PROBE %s %d" % (name, ok), flush=True)


if __name__ == "__main__":
    main()
