"""Drop-in alias: lets the reference's main.py (``from multigrid import ...``) resolve to the MI355X build."""
from deflatedmlmc_schwinger_amd.multigrid import *  # noqa: F401,F403
