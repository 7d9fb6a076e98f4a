"""CPU oracle (test infrastructure only) -- see rogue_oracle.h."""
