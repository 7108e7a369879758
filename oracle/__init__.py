"""CPU oracle of the hot path: TEST INFRASTRUCTURE ONLY (see oracle/ojf_oracle.c header)."""
