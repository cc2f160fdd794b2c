"""Parts of bench.py (round 6: the harness split by concern; bench.py keeps the argument parser, the headline line and the
CPU-baseline leg -- the only place outside tests/ that touches oracle/)."""
