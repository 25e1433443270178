"""Loaders of the CPU checkers (test infrastructure, never part of the product):
oracle/liboracle.so, the CPU restatement of the reference's forward pass, and
oracle/_ref/libref.so, the reference's own sources compiled in place where
/root/reference exists.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module."""
import os

from nvllm_amd import capi

REPO_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(REPO_DIR, "oracle", "liboracle.so")
REF_DIR = os.path.join(REPO_DIR, "oracle", "_ref")
REF_LIB = os.path.join(REF_DIR, "libref.so")


def load_oracle() -> capi.Lib:
    return capi.Lib(ORACLE_LIB, "oracle")


def load_ref() -> capi.Lib:
    return capi.Lib(REF_LIB, "ref")
