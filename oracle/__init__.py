"""Test infrastructure only: CPU restatements of the reference's algorithms for the hot path.
Nothing under wtracker_amd/ may import this package."""
