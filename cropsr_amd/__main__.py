"""`python -m cropsr_amd` == the reference's `python3 CROPSR.py` on the MI355X engine."""
from .cli import main

main()
