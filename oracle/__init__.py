"""CPU oracle of the STA hot path - TEST INFRASTRUCTURE ONLY (see oracle/sta_oracle.py)."""
