"""Test / benchmark infrastructure that is NOT part of the product package `vima_amd`: seeded synthetic weights and inputs
(`vima_testing.synthetic`) shared by tests/, bench.py, oracle/make_golden.py, examples/ and scripts/. Nothing under vima_amd/
imports this package."""
