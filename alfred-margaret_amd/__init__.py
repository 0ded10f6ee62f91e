"""alfred-margaret_amd: MI355X-native Aho-Corasick hot path behind alfred-margaret's API.

Import as `alfred_margaret_amd` (see ../alfred_margaret_amd/__init__.py).  Layout:
  csrc/   HIP kernels + flattener + C ABI (include/am.h)      -> lib/libam.so
  host/   C++ mirror of Automaton / Searcher / Replacer        -> lib/libam_host.so
  api.py  ctypes front-end used by tests and bench.py
"""
