"""Empty shim (see ../README.md)."""
