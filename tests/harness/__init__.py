"""Test infrastructure: the stand-in driver for controllers where the reference's Simulator is absent (sim_harness.py)."""
