"""Drop-in counterparts of the reference's lib/networks packages (hot path on HIP kernels)."""
