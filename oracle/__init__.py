"""CPU oracle (test infrastructure only — see oracle/hotpath.py).

Nothing under ironcortex_amd/ may import this package.
"""
