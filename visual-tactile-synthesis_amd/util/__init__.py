"""Host-side helpers (mirror of the reference's `util` package, hot-path subset only)."""
