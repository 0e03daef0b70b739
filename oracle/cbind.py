"""Build + load oracle/c/libacg_oracle.so through the product's ctypes signature table.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The C oracle exports the ABI of
include/acgan_hip.h on HOST pointers, so tests can (a) cross-check it against the torch
restatement and (b) inject it as a stand-in device library for CPU-only host-logic tests.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, 'c', 'libacg_oracle.so')
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', os.path.join(_HERE, 'c')])
    return SO_PATH


def load():
    global _LIB
    if _LIB is None:
        build()
        from action_conditioned_gans_amd._lib import Library
        _LIB = Library(SO_PATH)
    return _LIB
