"""Drop-in alias: lets the reference's main.py (``from gateway import ...``) resolve to the MI355X build."""
from deflatedmlmc_schwinger_amd.gateway import *  # noqa: F401,F403
