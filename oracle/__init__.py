"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see ctc_numpy.py header).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from ctc_amd/.
"""
