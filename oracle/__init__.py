"""CPU oracle (test infrastructure only) -- see oracle/grspec.py header."""
