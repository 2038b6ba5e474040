"""Drop-in alias: lets the reference's main.py (``from utils import ...``) resolve to the MI355X build."""
from deflatedmlmc_schwinger_amd.utils import *  # noqa: F401,F403
