"""coati_amd: MI355X-native implementation of COATi's marginal pairwise alignment
hot path (Viterbi fill + traceback, Forward fill + stochastic traceback).

The product is the C-ABI shared library built from coati_amd/csrc (see
include/coati_hip.h) plus the C++ host layer under coati_amd/host that mirrors
libcoati's marginal API.  This Python package is only the plumbing the tests
and bench.py use to reach them.
"""
__all__ = ["hip", "host"]
