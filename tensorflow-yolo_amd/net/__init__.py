"""Network definitions and runtime of the HIP backend (mirror of the reference's `net` package)."""
