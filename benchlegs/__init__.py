"""Secondary legs of bench.py, one module per leg (the headline step stays in bench.py)."""
