"""Drop-in alias: lets the reference's main.py (``from examples import ...``) resolve to the MI355X build."""
from deflatedmlmc_schwinger_amd.examples import *  # noqa: F401,F403
