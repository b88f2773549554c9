"""Test-only torch restatements of the reference modules (float64 / float32 CPU comparators).  Not imported by the product."""
