"""Imports the product package (directory `radix-sort_amd/`, module name radix_sort_amd)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_package():
    import __graft_entry__ as entry
    return entry.load_package()
