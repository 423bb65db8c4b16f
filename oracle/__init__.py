"""CPU oracle of the CROPSR PAM-scan/score path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (cropsr_amd/) never does.
"""
