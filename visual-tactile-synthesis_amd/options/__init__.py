"""Option parsing for the hot-path models (mirror of the reference `options` package)."""
