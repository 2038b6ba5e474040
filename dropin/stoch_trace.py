"""Drop-in alias: lets the reference's main.py (``from stoch_trace import ...``) resolve to the MI355X build."""
from deflatedmlmc_schwinger_amd.stoch_trace import *  # noqa: F401,F403
