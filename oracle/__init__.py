"""Test-only CPU oracle (see gpcsd_oracle.py header).  Never imported by gpcsd_amd."""
