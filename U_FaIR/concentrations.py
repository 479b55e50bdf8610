"""The reference's import path, kept so that its callers need no edit:

    from U_FaIR.concentrations import calculate_hfc_conc        # stujen/fiveEqSCM @ v0, tests/unit/test_hfcs.py:3

The reference declares `U_FaIR` as its package (setup.py:36) without an `__init__.py`; so does this
directory.  Nothing is implemented here: the name is re-exported from the drop-in module
`fiveeqscm_amd.concentrations`, whose `calculate_hfc_conc` keeps the reference's signature and behaviour
(U_FaIR/concentrations.py:4-5 of the reference) and whose other entry points (`run_ensemble`,
`calculate_hfc_conc_ensemble`) are this build's additions on the MI355X.
"""
from fiveeqscm_amd.concentrations import (calculate_hfc_conc, calculate_hfc_conc_ensemble,  # noqa: F401
                                          run_ensemble)

__all__ = ["calculate_hfc_conc", "calculate_hfc_conc_ensemble", "run_ensemble"]
