"""CPU oracle (test infrastructure only) -- see xm_oracle.c for scope and citations."""
