"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/kzg_oracle.c)."""
