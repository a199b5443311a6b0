"""Experiment switches of the engine.  NOT part of the product path: buctd_amd/ops.py imports this module only when the
environment says BUCTD_TUNING=1 (the A/B scripts under scratch/ do); then every switch of ops._SW can be overridden by the
variable BUCTD_<NAME>, e.g. BUCTD_TUNING=1 BUCTD_BRANCH_MAX=1 python bench.py."""
import os


def override(switches):
    for name in list(switches):
        v = os.environ.get("BUCTD_" + name)
        if v is not None:
            switches[name] = v
