"""Host-side mirrors of the reference's `utils` package for the supernet-training hot path."""
