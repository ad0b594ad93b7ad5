"""Parts of bench.py (the driver's contract lives in bench.py itself: CLI, rank launch, sub-results, the printed line)."""
