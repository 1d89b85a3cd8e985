"""Measurement scaffolding behind bench.py (the driver's entry point stays `python bench.py`): one module per concern."""
