"""Host-side mirrors of the reference's Evaluation/ scripts (config 1 of BASELINE.json) on the HIP engine."""
