"""The measurement legs bench.py reports beside its timed region (roofline of the timed kernel, copy rates, the beyond-the-cache
leg, the host side of a step) and the control plane of a multi-rank run.  bench.py itself keeps what the contract is about:
the arguments, the clocked K-step blocks, the one JSON line and the end-of-run exchange."""
