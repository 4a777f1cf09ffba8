"""The measurement legs of bench.py (one module per group of legs; bench.py itself holds the headline path and the JSON line)."""
