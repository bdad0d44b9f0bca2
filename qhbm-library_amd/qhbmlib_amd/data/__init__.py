"""Quantum data sources (reference: qhbmlib/data/__init__.py:20-23)."""
from qhbmlib_amd.data.quantum_data import QHBMData, QuantumData

__all__ = ["QHBMData", "QuantumData"]
