"""CPU oracle package -- test infrastructure only (see graphlily_oracle.c)."""
