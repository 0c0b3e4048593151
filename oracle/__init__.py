"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/qip_oracle.c). Not imported by rustqip_amd."""
