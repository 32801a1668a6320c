"""Test infrastructure only (see oracle/msda_oracle.c). Never imported by memotr_amd/."""
