"""Measurement and tuning scripts (see tools/README.md); not part of the product path."""
