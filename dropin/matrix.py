"""Drop-in alias: lets the reference's main.py (``from matrix import ...``) resolve to the MI355X build."""
from deflatedmlmc_schwinger_amd.matrix import *  # noqa: F401,F403
