"""The legs of bench.py, one module per leg (bench.py itself is the timed region and the line)."""
